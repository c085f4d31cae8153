// Probe: HBM read rate of the pointwise-conv tile access pattern: a block reads R rows (stride HW elements) x PIECE bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PIECE_B, int LOADS>
__global__ void __launch_bounds__(256) probe(const uint4* __restrict__ x, uint4* __restrict__ out, int rows_total, long long row_stride16,
                                             int pieces_per_row) {
    // block b: piece column (b % pieces_per_row), row group (b / pieces_per_row); lanes: PIECE_B/16 lanes per row
    constexpr int LPR = PIECE_B / 16;
    const int tid = threadIdx.x;
    const int pc = blockIdx.x % pieces_per_row, rg = blockIdx.x / pieces_per_row;
    constexpr int ROWS_PER_IT = 256 / LPR;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 v[LOADS];
#pragma unroll
    for (int it = 0; it < LOADS; ++it) {
        const int r = rg * (ROWS_PER_IT * LOADS) + it * ROWS_PER_IT + tid / LPR;
        v[it] = x[(long long)r * row_stride16 + (long long)pc * LPR + (tid % LPR)];
    }
#pragma unroll
    for (int it = 0; it < LOADS; ++it) { acc.x ^= v[it].x; acc.y ^= v[it].y; acc.z ^= v[it].z; acc.w ^= v[it].w; }
    out[(long long)blockIdx.x * 256 + tid] = acc;
}
int main() {
    const long long bytes = 50331648LL;   // 16 x 384 x 4096 x 2
    uint4 *x, *out;
    hipMalloc(&x, bytes); hipMalloc(&out, 64 << 20);
    hipMemset(x, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rows_total = 16 * 384;           // rows of 8 KB (4096 px bf16)
    const long long stride16 = 8192 / 16;
#define RUN(PIECE, LOADS, name)                                                                      \
    {                                                                                                \
        const int ppr = 8192 / PIECE;                                                                \
        const int rows_per_block = (256 / (PIECE / 16)) * LOADS;                                     \
        const int nblk = rows_total / rows_per_block * ppr;                                          \
        const long long max_x = (long long)(rows_total - 1) * stride16 + (long long)(ppr - 1) * (PIECE / 16) + (PIECE / 16 - 1); \
        if (PIECE / 16 > 256 || rows_per_block <= 0 || rows_total % rows_per_block != 0 || max_x >= bytes / 16 ||   \
            (long long)nblk * 256 > (64LL << 20) / 16) { printf("bad config %s\n", name); return 1; }              \
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<PIECE, LOADS>), dim3(nblk), dim3(256), 0, 0, x, out, rows_total, stride16, ppr); \
        hipEventRecord(e0);                                                                          \
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<PIECE, LOADS>), dim3(nblk), dim3(256), 0, 0, x, out, rows_total, stride16, ppr); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                 \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                  \
        fflush(stdout); printf("%-44s blocks=%5d  %.1f us  %.2f TB/s\n", name, nblk, ms * 50.0, bytes / (ms / 20 * 1e-3) / 1e12); \
    }
    RUN(256, 24, "piece 256 B x 384 rows/block (fan-in now)");
    RUN(256, 4, "piece 256 B x 64 rows/block");
    RUN(512, 24, "piece 512 B x 192 rows/block");
    RUN(512, 12, "piece 512 B x 96 rows/block");
    RUN(1024, 24, "piece 1 KB x 96 rows/block");
    RUN(1024, 6, "piece 1 KB x 24 rows/block");
    RUN(4096, 24, "piece 4 KB x 24 rows/block");
    RUN(4096, 4, "piece 4 KB x 4 rows/block");
    RUN(128, 24, "piece 128 B x 768 rows/block");
    return 0;
}
