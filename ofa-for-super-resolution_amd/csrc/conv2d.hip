// conv2d.hip -- dense KxK (3x3 / 5x5) convolution of the static ConvLayers as an implicit GEMM on the gfx950
// matrix cores: forward and input gradient (16-bit activations).
//
// Replaces nn.Conv2d in ConvLayer (reference ofa/layers.py:131-151; stem, residual convs, the two
// conv -> BN -> PixelShuffle up-sampling blocks and the output conv of OFAMobileNetS4, ofa_mbs4.py:65,105,120,123).
// These five layers are 68 % of the step's FLOPs at 4x; MIOpen runs them as NHWC iGEMM
// kernels wrapped in NCHW<->NHWC transposes at ~9 % of the bf16 MFMA peak.
//
//   Y[n, m, h, w] = sum_{tap=(ty,tx)} sum_{k} Wimg[tap][m][k] * X[n, k, h+ty-pad, w+tx-pad]
//
// forward: (m, k) = (co, ci); input gradient: the same kernel on dY with (m, k) = (ci, co) and the 180-degree
// rotated taps.  Roofline: MFMA (K = 25*Cin = 1600 for the 5x5 64-channel layers => AI >> ridge).
//
// Block = 2 output rows x 64 columns of one image x one slab of output channels.
//   * the (2+K-1) x (64+K-1) input window of a 64-channel chunk is staged ONCE into LDS, transposed to
//     [pixel][channel] so a B fragment (8 channels of one pixel) is one ds_read_b128; pixels of a row are stored
//     in 4 column-phase segments (col & 3), so the lanes of a fragment read -- which own columns 4c+t+tx -- touch
//     consecutive 128-byte records and the XOR swizzle of the 16-byte chunks makes the read conflict-free;
//   * weights come from a pre-swizzled bf16 image [tap][slab][chunk] (built per call by conv_prep_kernel from
//     the fp32 master weights, 0.8 MB for 5x5x64x256, L2-resident) and are copied tile by tile into LDS;
//   * MFMA column c of sub-tile t is the pixel (row c>>4, column 4*(c&15)+t): a lane owns 4 adjacent output
//     pixels and stores them with one 8-byte access.
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float cv_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 cv_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 cv_f16x8;
typedef __attribute__((ext_vector_type(8))) short cv_s16x8;

constexpr int CV_THREADS = 256;
constexpr int CV_TW = 64;     // output tile width
constexpr int CV_TH = 2;      // output tile height
constexpr int CV_SEG = 20;    // positions per column-phase segment (>= ceil(68 / 4); 4*SEG % 16 == 0 keeps the two
constexpr int CV_RP = 4 * CV_SEG;   // tile rows of a ds_read_b128 lane group on disjoint bank slots)

template <typename T> struct CvMma;
template <> struct CvMma<bf16_t> {
    static __device__ __forceinline__ cv_f32x16 run(cv_s16x8 a, cv_s16x8 b, cv_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cv_bf16x8, a), __builtin_bit_cast(cv_bf16x8, b),
                                                       c, 0, 0, 0);
    }
};
template <> struct CvMma<f16_t> {
    static __device__ __forceinline__ cv_f32x16 run(cv_s16x8 a, cv_s16x8 b, cv_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cv_f16x8, a), __builtin_bit_cast(cv_f16x8, b), c,
                                                      0, 0, 0);
    }
};

// [rows][64 k] 16-bit operand tile: 128-byte rows, 16-byte chunks XOR-swizzled (same format as pwconv.hip)
__device__ __host__ __forceinline__ int cv_tile_off(int row, int k) {
    return row * 128 + ((((k >> 3) ^ ((row >> 1) & 7)) << 4)) + (k & 7) * 2;
}
__device__ __forceinline__ int cv_tile_chunk(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// LDS record index of input-window pixel (r, col): rows of 68 records, 4 column-phase segments per row
__device__ __forceinline__ int cv_pos(int r, int col) { return r * CV_RP + (col & 3) * CV_SEG + (col >> 2); }

// wave decomposition: MODE 0: 128-row slab, wave = row block, 4 sub-tiles;  1: 64 rows, wave = (row block, pixel half);
// 2: 32 rows, wave = pixel quarter
template <int MODE> struct CvMode;
template <> struct CvMode<0> { static constexpr int ROWS = 128, NSUB = 4; };
template <> struct CvMode<1> { static constexpr int ROWS = 64, NSUB = 2; };
template <> struct CvMode<2> { static constexpr int ROWS = 32, NSUB = 1; };

// ---- weight image: [tap][slab][kchunk] tiles of [ROWS][64] in the swizzled operand format
template <typename T>
__global__ void __launch_bounds__(256) conv_prep_kernel(const float* __restrict__ w, T* __restrict__ wimg, int Cout,
                                                        int Cin, int KS, int dgrad, int M, int Kdim, int rows, int nslab,
                                                        int nkc, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx & 63);
    long long t = idx >> 6;
    const int r = (int)(t % rows);
    t /= rows;
    const int kc = (int)(t % nkc);
    t /= nkc;
    const int slab = (int)(t % nslab);
    const int tap = (int)(t / nslab);
    const int m = slab * rows + r, kk = kc * 64 + k;
    float v = 0.f;
    if (m < M && kk < Kdim) {
        const int taps = KS * KS;
        v = dgrad ? w[((long long)kk * Cin + m) * taps + (taps - 1 - tap)]    // rows = ci, k = co, rotated tap
                  : w[((long long)m * Cin + kk) * taps + tap];                // rows = co, k = ci
    }
    const long long tile = ((long long)tap * nslab + slab) * nkc + kc;
    char* base = reinterpret_cast<char*>(wimg) + tile * (long long)rows * 128;
    *reinterpret_cast<uint16_t*>(base + cv_tile_off(r, k)) = from_float<T>(v).v;
}

template <typename T, int KS, int MODE>
__global__ void __launch_bounds__(CV_THREADS) conv_igemm_kernel(const T* __restrict__ x, const T* __restrict__ wimg,
                                                                T* __restrict__ y, int Cx, int M, int H, int W,
                                                                int tiles_x, int nkc) {
    constexpr int PAD = KS / 2;
    constexpr int RH = CV_TH + KS - 1;
    constexpr int RW = CV_TW + KS - 1;
    constexpr int ROWS = CvMode<MODE>::ROWS;
    constexpr int NSUB = CvMode<MODE>::NSUB;
    __shared__ __attribute__((aligned(16))) char Xt[RH * CV_RP * 128];
    __shared__ __attribute__((aligned(16))) char Wt[ROWS * 128];

    const int tile = blockIdx.x;
    const int ty0 = (tile / tiles_x) * CV_TH, tx0 = (tile % tiles_x) * CV_TW;
    const int n = blockIdx.y, slab = blockIdx.z, nslab = gridDim.z;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int cb = MODE == 0 ? wave : (MODE == 1 ? (wave & 1) : 0);
    const int sub0 = MODE == 0 ? 0 : (MODE == 1 ? 2 * (wave >> 1) : wave);
    const int prow = c >> 4, pcol = 4 * (c & 15);   // this lane's pixel inside the tile (column of sub-tile 0)

    cv_f32x16 acc[NSUB];
#pragma unroll
    for (int t = 0; t < NSUB; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const T* xn = x + (long long)n * Cx * H * W;
    for (int kc = 0; kc < nkc; ++kc) {
        __syncthreads();   // previous chunk's readers are done with Xt / Wt
        // ---- stage the input window of channels [64kc, 64kc+64): aligned 16-byte runs of 8 columns, transposed
        //      into [pixel][channel] records
        constexpr int NCH = (CV_TW + 16) / 8;   // 10 runs cover columns [tx0-8, tx0+72)
        constexpr int TOTAL = 64 * RH * NCH;
        constexpr int BATCH = 5;   // independent 16-byte loads in flight per thread, then their LDS scatter
#pragma unroll 1
        for (int q0 = 0; q0 < TOTAL; q0 += BATCH * CV_THREADS) {
            uint4 v[BATCH];
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int q = q0 + tid + it * CV_THREADS;
                const int ch = q % NCH;
                const int r = (q / NCH) % RH;
                const int ci = q / (NCH * RH);
                const int gy = ty0 - PAD + r;
                const int gx = tx0 - 8 + 8 * ch;
                v[it] = make_uint4(0, 0, 0, 0);
                if (q < TOTAL && 64 * kc + ci < Cx && gy >= 0 && gy < H && gx >= 0 && gx < W)
                    v[it] = *reinterpret_cast<const uint4*>(xn + ((long long)(64 * kc + ci) * H + gy) * W + gx);
            }
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int q = q0 + tid + it * CV_THREADS;
                if (q < TOTAL) {
                    const int ch = q % NCH;
                    const int r = (q / NCH) % RH;
                    const int ci = q / (NCH * RH);
                    const uint32_t wds[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
                    const int cisw = ci >> 3, cioff = (ci & 7) * 2;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int tcol = 8 * ch - 8 + PAD + i;   // column inside the staged window
                        if (tcol >= 0 && tcol < RW) {
                            const int P = cv_pos(r, tcol);
                            const uint16_t e = (uint16_t)(i & 1 ? (wds[i >> 1] >> 16) : (wds[i >> 1] & 0xffffu));
                            *reinterpret_cast<uint16_t*>(Xt + P * 128 + (((cisw ^ ((P >> 1) & 7)) << 4)) + cioff) = e;
                        }
                    }
                }
            }
        }
        const int nks = min(4, (Cx - 64 * kc + 15) >> 4);   // 16-channel k-steps that hold data
        // weight tiles: the next tap's tile is fetched into registers while the current tap computes
        constexpr int WIT = ROWS * 8 / CV_THREADS;   // uint4 per thread per tile (4 / 2 / 1)
        const char* wbase = reinterpret_cast<const char*>(wimg) + ((long long)slab * nkc + kc) * (ROWS * 128);
        const long long wtap = (long long)nslab * nkc * (ROWS * 128);
        uint4 wreg[WIT];
#pragma unroll
        for (int j = 0; j < WIT; ++j) wreg[j] = reinterpret_cast<const uint4*>(wbase)[tid + j * CV_THREADS];
        for (int tap = 0; tap < KS * KS; ++tap) {
            __syncthreads();   // Xt staged (first tap) / previous tap's Wt readers done
#pragma unroll
            for (int j = 0; j < WIT; ++j) reinterpret_cast<uint4*>(Wt)[tid + j * CV_THREADS] = wreg[j];
            __syncthreads();
            if (tap + 1 < KS * KS) {
#pragma unroll
                for (int j = 0; j < WIT; ++j)
                    wreg[j] = reinterpret_cast<const uint4*>(wbase + (tap + 1) * wtap)[tid + j * CV_THREADS];
            }
            const int ty = tap / KS, tx = tap - ty * KS;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < nks) {
                    const cv_s16x8 af = *reinterpret_cast<const cv_s16x8*>(Wt + cv_tile_chunk(32 * cb + c, 2 * s + h));
#pragma unroll
                    for (int t = 0; t < NSUB; ++t) {
                        const int P = cv_pos(prow + ty, pcol + sub0 + t + tx);
                        const cv_s16x8 bf =
                            *reinterpret_cast<const cv_s16x8*>(Xt + P * 128 + (((2 * s + h) ^ ((P >> 1) & 7)) << 4));
                        acc[t] = CvMma<T>::run(af, bf, acc[t]);
                    }
                }
            }
        }
    }
    // ---- epilogue: lane owns NSUB adjacent pixels of row (ty0 + prow) for 16 output channels
    const int oy = ty0 + prow, ox = tx0 + pcol + sub0;
    if (oy < H && ox < W) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int m = slab * ROWS + 32 * cb + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (m < M) {
                T* dst = y + (((long long)n * M + m) * H + oy) * W + ox;
                if constexpr (NSUB == 4) {
                    *reinterpret_cast<uint2*>(dst) =
                        make_uint2(pack2<T>(acc[0][reg], acc[1][reg]), pack2<T>(acc[2][reg], acc[3][reg]));
                } else if constexpr (NSUB == 2) {
                    *reinterpret_cast<uint32_t*>(dst) = pack2<T>(acc[0][reg], acc[1][reg]);
                } else {
                    *dst = from_float<T>(acc[0][reg]);
                }
            }
        }
    }
}

struct CvPlan {
    bool ok;
    int mode, rows, nslab, nkc, M, Kdim;
    size_t img_bytes;
};

static CvPlan cv_plan(int64_t Cin, int64_t Cout, int K, int dgrad) {
    CvPlan p{};
    p.M = (int)(dgrad ? Cin : Cout);
    p.Kdim = (int)(dgrad ? Cout : Cin);
    p.ok = (K == 3 || K == 5) && p.M > 0 && p.Kdim > 0;
    p.mode = p.M > 64 ? 0 : (p.M > 32 ? 1 : 2);
    p.rows = p.mode == 0 ? 128 : (p.mode == 1 ? 64 : 32);
    p.nslab = (int)cdiv(p.M, p.rows);
    p.nkc = (int)cdiv(p.Kdim, 64);
    p.img_bytes = (size_t)K * K * p.nslab * p.nkc * p.rows * 128;
    return p;
}

template <typename T>
static int launch_conv2d(const char* name, const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout,
                         int64_t H, int64_t W, int K, int dgrad, void* ws, hipStream_t st) {
    const CvPlan p = cv_plan(Cin, Cout, K, dgrad);
    const long long total = (long long)K * K * p.nslab * p.nkc * p.rows * 64;
    hipLaunchKernelGGL((conv_prep_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w, (T*)ws, (int)Cout,
                       (int)Cin, K, dgrad, p.M, p.Kdim, p.rows, p.nslab, p.nkc, total);
    int rc = check_launch(name);
    if (rc) return rc;
    const int tiles_x = (int)cdiv(W, CV_TW), tiles_y = (int)cdiv(H, CV_TH);
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)N, (unsigned)p.nslab);
#define OFASR_CV(KS, MODE)                                                                                         \
    hipLaunchKernelGGL((conv_igemm_kernel<T, KS, MODE>), grid, dim3(CV_THREADS), 0, st, (const T*)x, (const T*)ws,  \
                       (T*)y, p.Kdim, p.M, (int)H, (int)W, tiles_x, p.nkc)
    if (K == 5) {
        if (p.mode == 0) OFASR_CV(5, 0); else if (p.mode == 1) OFASR_CV(5, 1); else OFASR_CV(5, 2);
    } else {
        if (p.mode == 0) OFASR_CV(3, 0); else if (p.mode == 1) OFASR_CV(3, 1); else OFASR_CV(3, 2);
    }
#undef OFASR_CV
    return check_launch(name);
}

static int conv2d_entry(const char* name, const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout,
                        int64_t H, int64_t W, int K, int dtype, int dgrad, void* ws, size_t ws_bytes, void* stream) {
    OFASR_REQUIRE(x && w && y, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED,
                  "%s: 16-bit activations only (fp32 runs on the vendor library)", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(W % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: needs W %% 8 == 0 and 16-byte aligned tensors", name);
    OFASR_REQUIRE(N <= 65535 && H * W <= (1LL << 31), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    const CvPlan p = cv_plan(Cin, Cout, K, dgrad);
    OFASR_REQUIRE(ws && ws_bytes >= p.img_bytes, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  ws_bytes, p.img_bytes);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_BF16) return launch_conv2d<bf16_t>(name, x, w, y, N, Cin, Cout, H, W, K, dgrad, ws, st);
    return launch_conv2d<f16_t>(name, x, w, y, N, Cin, Cout, H, W, K, dgrad, ws, st);
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT size_t ofasr_conv2d_workspace(int64_t Cin, int64_t Cout, int K, int dgrad) {
    if (Cin <= 0 || Cout <= 0 || !(K == 3 || K == 5)) return 0;
    return cv_plan(Cin, Cout, K, dgrad).img_bytes;
}

OFASR_EXPORT int ofasr_conv2d_fwd(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                  int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_entry("ofasr_conv2d_fwd", x, w, y, N, Cin, Cout, H, W, K, dtype, 0, workspace, workspace_bytes, stream);
}

OFASR_EXPORT int ofasr_conv2d_dgrad(const void* dy, const float* w, void* dx, int64_t N, int64_t Cin, int64_t Cout,
                                    int64_t H, int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return conv2d_entry("ofasr_conv2d_dgrad", dy, w, dx, N, Cin, Cout, H, W, K, dtype, 1, workspace, workspace_bytes,
                        stream);
}
