// What do the start / stop events of hipExtLaunchKernel measure?  Launch A (long) then B (short) back to back on one
// stream, compare  elapsed(startB, stopB), elapsed(stopA, stopB), elapsed(startA, stopA)  with the kernels' own
// wall-clock (s_memrealtime, 100 MHz) first-wave-start .. last-wave-end span.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin(unsigned long long* span, int iters, float* sink) {
    unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) *sink = a;
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(&span[0], t0);
        atomicMax(&span[1], t1);
    }
}
int main() {
    unsigned long long *sa, *sb;
    float* sink;
    hipMalloc(&sa, 16); hipMalloc(&sb, 16); hipMalloc(&sink, 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a0, a1, b0, b1, r0, r1;
    hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1); hipEventCreate(&r0); hipEventCreate(&r1);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long init[2] = {~0ull, 0ull};
        hipMemcpy(sa, init, 16, hipMemcpyHostToDevice); hipMemcpy(sb, init, 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, a0, a1, 0, sa, 40000, sink);
        hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, b0, b1, 0, sb, 4000, sink);
        hipDeviceSynchronize();
        unsigned long long ha[2], hb[2];
        hipMemcpy(ha, sa, 16, hipMemcpyDeviceToHost); hipMemcpy(hb, sb, 16, hipMemcpyDeviceToHost);
        float eA, eB, eAB;
        hipEventElapsedTime(&eA, a0, a1); hipEventElapsedTime(&eB, b0, b1);
        hipError_t e = hipEventElapsedTime(&eAB, a1, b1);
        printf("rep %d: in-kernel A %.1f us  B %.1f us | ext events A %.1f us  B %.1f us | stopA->stopB %.1f us (%s)\n", rep,
               (ha[1] - ha[0]) / 100.0, (hb[1] - hb[0]) / 100.0, eA * 1e3, eB * 1e3, eAB * 1e3, hipGetErrorString(e));
        // the round-2 scheme: hipEventRecord brackets
        hipMemcpy(sb, init, 16, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, sa, 40000, sink);
        hipEventRecord(r0, st);
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, sb, 4000, sink);
        hipEventRecord(r1, st);
        hipDeviceSynchronize();
        hipMemcpy(hb, sb, 16, hipMemcpyDeviceToHost);
        float eR; hipEventElapsedTime(&eR, r0, r1);
        printf("        record brackets B %.1f us (in-kernel %.1f)\n", eR * 1e3, (hb[1] - hb[0]) / 100.0);
    }
    // (i) an idle queue: B alone; (ii) plain launch of A, then B with events; (iii) B with events on stream 2 while A runs on stream 1
    hipStream_t st2; hipStreamCreate(&st2);
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long init[2] = {~0ull, 0ull}, hb[2];
        float e;
        hipMemcpy(sb, init, 16, hipMemcpyHostToDevice); hipDeviceSynchronize();
        hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, b0, b1, 0, sb, 600, sink);
        hipDeviceSynchronize(); hipMemcpy(hb, sb, 16, hipMemcpyDeviceToHost); hipEventElapsedTime(&e, b0, b1);
        printf("idle queue      : in-kernel %.1f us, ext events %.1f us\n", (hb[1] - hb[0]) / 100.0, e * 1e3);
        hipMemcpy(sb, init, 16, hipMemcpyHostToDevice); hipDeviceSynchronize();
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, sa, 4000, sink);
        hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, b0, b1, 0, sb, 600, sink);
        hipDeviceSynchronize(); hipMemcpy(hb, sb, 16, hipMemcpyDeviceToHost); hipEventElapsedTime(&e, b0, b1);
        printf("after plain A   : in-kernel %.1f us, ext events %.1f us\n", (hb[1] - hb[0]) / 100.0, e * 1e3);
        hipMemcpy(sb, init, 16, hipMemcpyHostToDevice); hipDeviceSynchronize();
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, sa, 40000, sink);
        hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st2, b0, b1, 0, sb, 600, sink);
        hipDeviceSynchronize(); hipMemcpy(hb, sb, 16, hipMemcpyDeviceToHost); hipEventElapsedTime(&e, b0, b1);
        printf("beside A (st2)  : in-kernel %.1f us, ext events %.1f us\n", (hb[1] - hb[0]) / 100.0, e * 1e3);
        // a burst of 20 short kernels with events, back to back
        hipEvent_t ev[40]; for (int i = 0; i < 40; ++i) hipEventCreate(&ev[i]);
        hipDeviceSynchronize();
        for (int i = 0; i < 20; ++i) hipExtLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, st, ev[2 * i], ev[2 * i + 1], 0, sa, 600, sink);
        hipDeviceSynchronize();
        float tot = 0, span; for (int i = 0; i < 20; ++i) { hipEventElapsedTime(&e, ev[2 * i], ev[2 * i + 1]); tot += e; }
        hipEventElapsedTime(&span, ev[0], ev[39]);
        printf("burst of 20     : mean ext events %.1f us, first start -> last end / 20 = %.1f us\n", tot * 1e3 / 20, span * 1e3 / 20);
    }
    return 0;
}
