// valu_rate.hip -- issue rate of the VALU forms a depthwise inner loop can be built from (gfx950):
//   v_fma_f32, v_pk_fma_f32, v_dot2c_f32_bf16, v_dot2c_f32_f16;  8 waves per SIMD-pair... (grid fills the chip)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o /tmp/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) float f2;

template <int MODE>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ in, float* __restrict__ out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 8 + i) & 4095]; b[i] = in[(t * 8 + i + 17) & 4095]; }
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned x = a[(i + r) & 7], y = b[(i * 3 + r) & 7];
                if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
                else if constexpr (MODE == 2) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x), __builtin_bit_cast(bf2, y), acc[i], false);
                else if constexpr (MODE == 3) acc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, x), __builtin_bit_cast(h2, y), acc[i], false);
            }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f2 xx = {__uint_as_float(a[(i + r) & 7]), __uint_as_float(a[(i + r + 1) & 7])};
                    f2 yy = {__uint_as_float(b[(i * 3 + r) & 7]), __uint_as_float(b[(i * 3 + r + 1) & 7])};
                    f2 cc = {acc[i], acc[i + 1]};
                    cc = __builtin_elementwise_fma(xx, yy, cc);
                    acc[i] = cc.x;
                    acc[i + 1] = cc.y;
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[t] = s;
}

template <int MODE> static void run(const char* name, const unsigned* in, float* out, double ops_per_inst) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 /*waves*/ * iters * 64.0 * (MODE == 1 ? 0.5 : 1.0);   // wave-instructions
    const double per_simd = inst / (256.0 * 4.0);
    printf("%-22s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz, %.1f Tops/s\n", name, ms,
           ms * 1e-3 * 2.4e9 / per_simd, inst * 64 * ops_per_inst / (ms * 1e-3) / 1e12);
}

int main() {
    unsigned* in; float* out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemset(in, 0x3c, 4096 * 4);
    run<0>("v_fma_f32", in, out, 2);
    run<1>("v_pk_fma_f32", in, out, 4);
    run<2>("v_dot2c_f32_bf16", in, out, 4);
    run<3>("v_dot2c_f32_f16", in, out, 4);
    return 0;
}
