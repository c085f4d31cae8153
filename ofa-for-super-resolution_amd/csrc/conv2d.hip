// conv2d.hip -- dense KxK (3x3 / 5x5) convolution of the static ConvLayers as an implicit GEMM on the gfx950
// matrix cores: forward and input gradient (16-bit activations).
//
// Replaces nn.Conv2d in ConvLayer (reference ofa/layers.py:131-151; stem, residual convs, the two
// conv -> BN -> PixelShuffle up-sampling blocks and the output conv of OFAMobileNetS4, ofa_mbs4.py:65,105,120,123).
// These five layers are 68 % of the step's FLOPs at 4x.
//
//   Y[n, m, h, w] = sum_{tap=(ty,tx)} sum_{k} Wimg[tap][m][k] * X[n, k, h+ty-pad, w+tx-pad]
//
// forward: (m, k) = (co, ci); input gradient: the same kernel on dY with (m, k) = (ci, co) and the 180-degree
// rotated taps.  Roofline: MFMA (K = 25*Cin = 1600 for the 5x5 64-channel layers => AI >> ridge).
//
// Block = (2*WP) output rows x 64 columns of one image x (64*WM or 32*WM) output channels; a wave owns RB 32-row
// blocks x 2 rows x 64 columns (4 MFMA sub-tiles: column c of sub-tile t is the pixel (row c>>4, column 4*(c&15)+t),
// so a lane owns 4 adjacent output pixels and stores them with one 8-byte access).
//   * B operand: the (TH+K-1) x (64+K-1) input window of a 32-channel chunk is staged ONCE into LDS transposed to
//     [pixel][32 channels] 64-byte records (8x8 register transposes, 16-byte LDS stores), so a B fragment (8
//     channels of one pixel) is one ds_read_b128.  Pixels of a window row sit in 4 column-phase segments (col & 3):
//     the lanes of a fragment read -- which own columns 4c+t+tx -- then touch consecutive records, and the XOR of
//     the 16-byte chunk index with (record>>2)&3 makes every 16-lane group of the read conflict-free.
//     For one (ty, k-step) the K*4 (tx, t) MFMAs of a row block need only the 8 fragments u = t+tx in [0, 8).
//   * A operand: straight from a fragment-ordered bf16 weight image (built per call by conv_prep_kernel from the
//     fp32 master weights; 0.8 MB for 5x5x64x256, L2-resident): one coalesced 16-byte load per lane per fragment,
//     re-issued one (ty, k-step) ahead into the registers the MFMAs just released.  No LDS, no barrier per tap.
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float cv_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 cv_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 cv_f16x8;
typedef __attribute__((ext_vector_type(8))) short cv_s16x8;

constexpr int CV_TW = 64;     // output tile width
constexpr int CV_SEG = 20;    // records per column-phase segment (>= ceil((64 + 4) / 4))
constexpr int CV_RP = 4 * CV_SEG;   // records per window row; % 16 == 0 keeps both rows of a lane group bank-disjoint
constexpr int CV_KC = 32;     // channels per staged chunk (64-byte records)

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

// LDS record index of window pixel (r, col), and the byte offset of its 8-channel group j (0..3)
__device__ __forceinline__ int cv_pos(int r, int col) { return r * CV_RP + (col & 3) * CV_SEG + (col >> 2); }
__device__ __forceinline__ int cv_rec(int P, int j) { return P * 64 + ((j ^ ((P >> 2) & 3)) << 4); }

// ---- weight image: 16-byte A fragments in consumption order [kc][ty][s][tx][row block][lane]
//      lane (r = lane&31, h = lane>>5) holds rows 32*rb + r, k = 32*kc + 16*s + 8*h + (0..7)
struct CvBn {
    float* ss;            // out: scale[M] | shift[M]; nullptr: no epilogue operands wanted
    const float* gamma;   // nullptr: no BN (scale 1, shift 0)
    const float* beta;
    const float* mean;
    const float* var;
    float eps;
};

template <typename T>
__global__ void __launch_bounds__(256) conv_prep_kernel(const float* __restrict__ w, T* __restrict__ wimg, int Cin,
                                                        int KS, int dgrad, int M, int Kdim, int nrb, long long total,
                                                        CvBn bn) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (bn.ss != nullptr && idx < M) {   // scale | shift of the eval-mode BN behind the conv (identity without one)
        float sc = 1.f, sh = 0.f;
        if (bn.gamma != nullptr) {
            sc = bn.gamma[idx] * rsqrtf(bn.var[idx] + bn.eps);
            sh = bn.beta[idx] - bn.mean[idx] * sc;
        }
        bn.ss[idx] = sc;
        bn.ss[M + idx] = sh;
    }
    if (idx >= total) return;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    long long t = idx >> 9;
    const int rb = (int)(t % nrb);
    t /= nrb;
    const int tx = (int)(t % KS);
    t /= KS;
    const int s = (int)(t & 1);
    t >>= 1;
    const int ty = (int)(t % KS);
    const int kc = (int)(t / KS);
    const int m = rb * 32 + (lane & 31), kk = kc * CV_KC + 16 * s + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (m < M && kk < Kdim) {
        const int taps = KS * KS, tap = ty * KS + tx;
        v = dgrad ? w[((long long)kk * Cin + m) * taps + (taps - 1 - tap)]    // rows = ci, k = co, rotated tap
                  : w[((long long)m * Cin + kk) * taps + tap];                // rows = co, k = ci
    }
    wimg[idx] = from_float<T>(v);
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's outstanding global loads and
// stores (vmcnt(0)), i.e. it would wait for the A fragments requested just before it instead of letting their L2 latency
// hide behind the staging of the window
__device__ __forceinline__ void cv_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// inference epilogue (ofasr_conv2d_infer_run): ss = scale[M] | shift[M] of the eval-mode BatchNorm that follows the conv
// (nullptr: plain conv output, the training path), act 0 none / 1 ReLU6 / 2 PixelShuffle(2) store ([N, M/4, 2H, 2W])
struct CvEpi {
    const float* ss;
    int act;
    float2* stat;   // training: per-channel (sum, sum of squares) of what each wave stores -> stat[m * P + unit], or nullptr
    int P;          //   unit = ((n * tiles + tile) * WP + wp): one per wave's 2 x 64 pixel strip
};

template <typename T, int KS, int RB, int WM, int WP, bool ONEK>
__global__ void __launch_bounds__(64 * WM * WP, 2) conv_igemm_kernel(const T* __restrict__ x, const T* __restrict__ wimg,
                                                                     T* __restrict__ y, int Cx, int M, int H, int W,
                                                                     int tiles_x, int nkc, int nrb, CvEpi epi) {
    constexpr int THREADS = 64 * WM * WP;
    constexpr int PAD = KS / 2;
    constexpr int TH = 2 * WP;
    constexpr int RH = TH + KS - 1;
    constexpr int RW = CV_TW + KS - 1;
    constexpr int NIT = 2 * KS;   // (ty, k-step) iterations per chunk
    __shared__ __attribute__((aligned(16))) char Xt[RH * CV_RP * 64];

    const int tile = blockIdx.x;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * CV_TW;
    const int n = blockIdx.y, slab = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave % WM, wp = wave / WM;
    const int c = lane & 31, h = lane >> 5;
    const int prow = 2 * wp + (c >> 4), pcol = 4 * (c & 15);   // this lane's pixel inside the tile (sub-tile 0)
    const int grb0 = (slab * WM + wm) * RB;                    // first global row block of this wave

    // B-fragment addresses: record P = P0 + D(ty, u) with the lane part P0 = prow*RP + (c&15) and the compile-time
    // D = ty*RP + (u&3)*SEG + (u>>2).  The swizzle key ((P>>2)&3) = (K0 + (D>>2) + carry)&3 with K0 = (P0>>2)&3 and
    // carry = ((c&3) == 3 && u >= 4): 2 (carry sets) x 4 (key deltas) x 2 (k-steps) base registers, the rest is the
    // instruction's immediate offset.
    const int P0 = prow * CV_RP + (c & 15);
    int bbase[2][4][2];
#pragma unroll
    for (int cs = 0; cs < 2; ++cs)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int key = (((P0 >> 2) & 3) + m + (cs && (c & 3) == 3 ? 1 : 0)) & 3;
                bbase[cs][m][ks] = P0 * 64 + (((2 * ks + h) ^ key) << 4);
            }
    auto read_b = [&](int ty, int u, int ks) -> cv_s16x8 {
        const int D = ty * CV_RP + (u & 3) * CV_SEG + (u >> 2);
        return *reinterpret_cast<const cv_s16x8*>(Xt + bbase[u >> 2][(D >> 2) & 3][ks] + D * 64);
    };

    cv_f32x16 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][t][i] = 0.f;

    const T* xn = x + (long long)n * Cx * H * W;
    const uint4* wbase = reinterpret_cast<const uint4*>(wimg);   // wave-uniform: fragment addresses stay in SGPRs
    for (int kc = 0; kc < nkc; ++kc) {
        // A fragments of iteration 0 (their latency hides behind the staging below)
        cv_s16x8 A[KS][RB];
        const long long it_stride = (long long)KS * nrb * 64;   // uint4 per (ty, s) iteration
        const uint4* wk = wbase + (long long)kc * NIT * it_stride + (long long)grb0 * 64;
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                A[tx][rb] = __builtin_bit_cast(cv_s16x8, (wk + ((long long)tx * nrb + rb) * 64)[lane]);

        cv_lds_barrier();   // previous chunk's readers are done with Xt (LDS only: the A requests above stay in flight)
        // ---- stage the window of channels [32kc, 32kc+32): task = (8-channel group, window row, 8-column run);
        //      8 x 16-byte loads, 8x8 16-bit transpose in registers, 8 x 16-byte record stores
        constexpr int NRUN = (CV_TW + 16) / 8;   // 10 runs cover columns [tx0-8, tx0+72)
        constexpr int NTASK = 4 * RH * NRUN;
#pragma unroll 1
        for (int q = tid; q < NTASK; q += THREADS) {
            const int run = q % NRUN;
            const int r = (q / NRUN) % RH;
            const int g = q / (NRUN * RH);
            const int gy = ty0 - PAD + r;
            const int gx = tx0 - 8 + 8 * run;
            const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
            const int cbase = CV_KC * kc + 8 * g;
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i] = make_uint4(0, 0, 0, 0);
                if (inb && cbase + i < Cx)
                    v[i] = *reinterpret_cast<const uint4*>(xn + ((long long)(cbase + i) * H + gy) * W + gx);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // dword k of every channel: pixels 2k, 2k+1
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t a = k == 0 ? v[2 * m].x : k == 1 ? v[2 * m].y : k == 2 ? v[2 * m].z : v[2 * m].w;
                    const uint32_t b = k == 0 ? v[2 * m + 1].x : k == 1 ? v[2 * m + 1].y : k == 2 ? v[2 * m + 1].z
                                                                                                  : v[2 * m + 1].w;
                    lo[m] = __builtin_amdgcn_perm(b, a, 0x05040100u);
                    hi[m] = __builtin_amdgcn_perm(b, a, 0x07060302u);
                }
                const int tc0 = 8 * run - 8 + PAD + 2 * k;   // window column of pixel 2k
                if (tc0 >= 0 && tc0 < RW)
                    *reinterpret_cast<uint4*>(Xt + cv_rec(cv_pos(r, tc0), g)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                if (tc0 + 1 >= 0 && tc0 + 1 < RW)
                    *reinterpret_cast<uint4*>(Xt + cv_rec(cv_pos(r, tc0 + 1), g)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            }
        }
        cv_lds_barrier();

        // (ty, k-step) iterations; ONEK (K <= 16: the 3-channel stem / head gradient) walks k-step 0 only
        constexpr int STEP = ONEK ? 2 : 1;
#pragma unroll
        for (int it = 0; it < NIT; it += STEP) {
            const int ty = it >> 1, ks = it & 1;
            cv_s16x8 B[8];
#pragma unroll
            for (int u = 0; u < KS + 3; ++u) B[u] = read_b(ty, u, ks);
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rb][t] = CvMma<T>::run(A[tx][rb], B[t + tx], acc[rb][t]);
                __builtin_amdgcn_sched_barrier(0);   // keep the refill below the MFMAs that free its registers
                if (it + STEP < NIT) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
                        A[tx][rb] = __builtin_bit_cast(
                            cv_s16x8, (wk + (it + STEP) * it_stride + ((long long)tx * nrb + rb) * 64)[lane]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- epilogue: lane owns 4 adjacent pixels of row (ty0 + prow) for 16 output channels per row block
    const int oy = ty0 + prow, ox = tx0 + pcol;
    if (epi.ss != nullptr) {   // inference: y = act(conv * scale[m] + shift[m]) in fp32, one rounding (workgroup-uniform)
        if (oy >= H || ox >= W) return;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int mb = (grb0 + rb) * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {          // registers 4q .. 4q+3 = channels mb + 8q + (0..3)
                float v[4][4];                     // [channel j][pixel t]
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = mb + 8 * q + j;
                    const int mc = m < M ? m : M - 1;
                    const float sc = epi.ss[mc], sh = epi.ss[M + mc];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float r = fmaf(acc[rb][t][4 * q + j], sc, sh);
                        if (epi.act == 1) r = fminf(fmaxf(r, 0.f), 6.f);
                        v[j][t] = r;
                    }
                }
                if (epi.act == 2) {
                    // PixelShuffle(2) as the store: channels 4C .. 4C+3 are the (dy, dx) = (j >> 1, j & 1) sub-pixels of output
                    // channel C; this lane holds all four for its 4 pixels -> two 16-byte rows of 8 output pixels
                    const int m0 = mb + 8 * q;
                    if (m0 < M) {
                        T* dst = y + (((long long)n * (M >> 2) + (m0 >> 2)) * (2 * H) + 2 * oy) * (2 * W) + 2 * ox;
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
                            *reinterpret_cast<uint4*>(dst + (long long)dy * (2 * W)) =
                                make_uint4(pack2<T>(v[2 * dy][0], v[2 * dy + 1][0]), pack2<T>(v[2 * dy][1], v[2 * dy + 1][1]),
                                           pack2<T>(v[2 * dy][2], v[2 * dy + 1][2]), pack2<T>(v[2 * dy][3], v[2 * dy + 1][3]));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mb + 8 * q + j;
                        if (m < M)
                            *reinterpret_cast<uint2*>(y + (((long long)n * M + m) * H + oy) * W + ox) =
                                make_uint2(pack2<T>(v[j][0], v[j][1]), pack2<T>(v[j][2], v[j][3]));
                    }
                }
            }
        }
        return;
    }
    const bool inside = oy < H && ox < W;
    if (inside) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = (grb0 + rb) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (m < M) {
                    T* dst = y + (((long long)n * M + m) * H + oy) * W + ox;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(pack2<T>(acc[rb][0][reg], acc[rb][1][reg]),
                                                                pack2<T>(acc[rb][2][reg], acc[rb][3][reg]));
                }
            }
        }
    }
    if (epi.stat != nullptr) {
        // BatchNorm statistics of the values as stored (ofa/layers.py:120-151: the BN that follows the conv): the 32 lanes
        // of a half-wave hold 128 pixels of the same 16 channels per row block -> reduce-scatter, one partial per
        // (channel, wave strip); pixels outside the image count as zero
        const int unit = (n * (int)gridDim.x + tile) * WP + wp;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float sv[16], qv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float v = inside ? to_float(from_float<T>(acc[rb][t][reg])) : 0.f;
                    ss += v;
                    qq = fmaf(v, v, qq);
                }
                sv[reg] = ss;
                qv[reg] = qq;
            }
            half_wave_row_sums(sv, qv, c);
            const int rho = half_wave_row_reg(c);
            const int m = (grb0 + rb) * 32 + (rho & 3) + 8 * (rho >> 2) + 4 * h;
            if ((c & 1) == 0 && m < M) epi.stat[(long long)m * epi.P + unit] = make_float2(sv[0], qv[0]);
        }
    }
}

struct CvPlan {
    bool ok;
    int cfg;        // 0: 256 rows/block (RB2 WM4 WP1); 1: 128 (RB2 WM2 WP2); 2: 64 (RB2 WM1 WP4); 3: 32 (RB1 WM1 WP4);
                    // 4: 64 rows as RB1 WM2 WP2 (4-row tiles: twice the blocks of cfg 2 for small images)
    int rows, th, threads, nslab, nrb, nkc, M, Kdim;
    size_t img_bytes;
};

static CvPlan cv_plan(int64_t Cin, int64_t Cout, int K, int dgrad, int64_t N = 0, int64_t H = 0, int64_t W = 0) {
    CvPlan p{};
    p.M = (int)(dgrad ? Cin : Cout);
    p.Kdim = (int)(dgrad ? Cout : Cin);
    p.ok = (K == 3 || K == 5) && p.M > 0 && p.Kdim > 0;
    p.cfg = p.M > 128 ? 0 : (p.M > 64 ? 1 : (p.M > 32 ? 2 : 3));
    // the weight image does not depend on the (N, H, W)-driven choice between cfg 2 and 4 (same 64-row slabs)
    if (p.cfg == 2 && N > 0 && N * cdiv(H, 8) * cdiv(W, CV_TW) < 512) p.cfg = 4;
    p.rows = p.cfg == 0 ? 256 : (p.cfg == 1 ? 128 : (p.cfg == 3 ? 32 : 64));
    p.th = p.cfg == 0 ? 2 : ((p.cfg == 1 || p.cfg == 4) ? 4 : 8);
    p.threads = 256;
    p.nslab = (int)cdiv(p.M, p.rows);
    p.nrb = p.nslab * (p.rows / 32);
    p.nkc = (int)cdiv(p.Kdim, CV_KC);
    p.img_bytes = (size_t)p.nkc * K * 2 * K * p.nrb * 1024;
    return p;
}

template <typename T>
static int launch_conv_prep(const char* name, const float* w, int64_t Cin, int64_t Cout, int K, int dgrad, void* ws, CvBn bn,
                            hipStream_t st) {
    const CvPlan p = cv_plan(Cin, Cout, K, dgrad);
    const long long total = (long long)(p.img_bytes / 2);
    OFASR_LAUNCH((conv_prep_kernel<T>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w, (T*)ws, (int)Cin, K,
                       dgrad, p.M, p.Kdim, p.nrb, total, bn);
    return check_launch(name);
}

template <typename T>
static int launch_conv_run(const char* name, const void* x, const void* ws, void* y, int64_t N, int64_t Cin, int64_t Cout,
                           int64_t H, int64_t W, int K, int dgrad, CvEpi epi, hipStream_t st) {
    const CvPlan p = cv_plan(Cin, Cout, K, dgrad, N, H, W);
    const int tiles_x = (int)cdiv(W, CV_TW), tiles_y = (int)cdiv(H, p.th);
    dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)N, (unsigned)p.nslab);
    prof_note((double)sizeof(T) * (double)N * (double)H * (double)W * (double)(Cin + Cout),
              2.0 * (double)N * (double)H * (double)W * (double)Cin * (double)Cout * K * K);
#define OFASR_CV(KS, RB, WM, WP, ONEK)                                                                                  \
    OFASR_LAUNCH((conv_igemm_kernel<T, KS, RB, WM, WP, ONEK>), grid, dim3(64 * WM * WP), 0, st, (const T*)x,         \
                       (const T*)ws, (T*)y, p.Kdim, p.M, (int)H, (int)W, tiles_x, p.nkc, p.nrb, epi)
#define OFASR_CVK(KS, ONEK)                                                                                         \
    switch (p.cfg) {                                                                                                \
        case 0: OFASR_CV(KS, 2, 4, 1, ONEK); break;                                                                 \
        case 1: OFASR_CV(KS, 2, 2, 2, ONEK); break;                                                                 \
        case 2: OFASR_CV(KS, 2, 1, 4, ONEK); break;                                                                 \
        case 4: OFASR_CV(KS, 1, 2, 2, ONEK); break;                                                                 \
        default: OFASR_CV(KS, 1, 1, 4, ONEK); break;                                                                \
    }
    const bool onek = p.Kdim <= 16;
    if (K == 5) {
        if (onek) { OFASR_CVK(5, true) } else { OFASR_CVK(5, false) }
    } else {
        if (onek) { OFASR_CVK(3, true) } else { OFASR_CVK(3, false) }
    }
#undef OFASR_CVK
#undef OFASR_CV
    return check_launch(name);
}

template <typename T>
static int launch_conv2d(const char* name, const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout,
                         int64_t H, int64_t W, int K, int dgrad, void* ws, hipStream_t st, StatOut so = StatOut{nullptr, 0}) {
    int rc = launch_conv_prep<T>(name, w, Cin, Cout, K, dgrad, ws, CvBn{}, st);
    if (rc) return rc;
    return launch_conv_run<T>(name, x, ws, y, N, Cin, Cout, H, W, K, dgrad, CvEpi{nullptr, 0, so.partial, so.P}, st);
}

// ================================================================================= weight gradient
//   dW[co, ci, ty, tx] = sum_{n, h, w} dY[n, co, h, w] * X[n, ci, h + ty - pad, w + tx - pad]
// GEMM with k = pixels.  A block owns (a slab of output channels) x (64 or 32 input channels) x ONE kernel row ty and
// a contiguous range of 2x64-pixel tiles (split-K); it keeps the K accumulators (tx = 0..K-1) of its row blocks in
// registers across the whole range and writes them once, in accumulator order, to its slab; conv_wgrad_reduce_kernel
// sums the slabs in a fixed order (deterministic) and scatters into the [Cout, Cin, K, K] fp32 gradient.
//   * A (dY, rows = co): the tile is copied as is ([co][128 pixels], 16-byte chunks XOR-swizzled with row&15) --
//     8 consecutive pixels of one channel are contiguous in NCHW, one ds_read_b128 per fragment;
//   * B (X, cols = ci): the 2 x 80 pixel window of kernel row ty is transposed on the way in to [pixel][ci] records
//     (8x8 register transposes, as the forward kernel) and read with ds_read_b64_tr_b16, so the tap shift tx is a
//     whole-record offset and needs no alignment.
// wave = (row-block group wco, 32-channel column block wci, k-step phase wk).
typedef __attribute__((ext_vector_type(4))) short cv_s16x4;
typedef __attribute__((address_space(3))) cv_s16x4 cv_lds_s16x4;

struct CvWgParams {
    int Cin, Cout, H, W, tiles_x, tiles_y, ntiles, ksplit, ncislab, nz;
};

template <int WCI> __device__ __forceinline__ int cvw_xoff(int R, int chunk) {   // byte offset of 16-byte chunk in record R
    if constexpr (WCI == 2) return R * 128 + ((((chunk >> 2) ^ ((R >> 1) & 1))) << 6) + ((chunk & 3) << 4);
    else return R * 64 + (chunk << 4);
}

template <typename T, int KS, int RBW, int WCO, int WCI, int WK>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            float* __restrict__ part, CvWgParams P) {
    static_assert(WCO * WCI * WK == 4, "4 waves");
    constexpr int PAD = KS / 2;
    constexpr int CIB = 32 * WCI;            // input channels per block
    constexpr int REC = 2 * CIB;             // bytes per pixel record
    constexpr int ROWS = 32 * RBW * WCO;     // output channels per block
    constexpr int XW = 80;                   // staged window columns [tx0-8, tx0+72)
    __shared__ __attribute__((aligned(16))) char Yt[ROWS * 256];
    __shared__ __attribute__((aligned(16))) char Xt[2 * XW * REC];

    // Block order: the K * nz blocks of one pixel split (kernel rows x channel slabs) read the same dY tiles and
    // overlapping X windows, so they should run at the same time on the same XCD (each XCD has its own L2; consecutive
    // workgroup ids go round the 8 XCDs): id = 8 * (G * q + group) + xcd with split = 8 q + xcd.  (With the split as
    // the fastest grid index the 10 blocks of a split sat on different XCDs and every one fetched its tiles from HBM:
    // 827 MB per launch for 168 MB of activations on 64 -> 256 @128x128.)
    const int G = KS * P.nz;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int group = jb % G, split = (jb / G) * 8 + xcd;
    if (split >= P.ksplit) return;     // workgroup-uniform: the grid is padded to whole rounds of 8 splits
    const int ty = group % KS, z = group / KS;
    const int coslab = z / P.ncislab, cislab = z % P.ncislab;
    const int tid = threadIdx.x, lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wco = wave % WCO, wci = (wave / WCO) % WCI, wk = wave / (WCO * WCI);
    const int c = lane & 31, h = lane >> 5;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;

    // B read bases: byte offset of this lane's 4 channels in record L, with the 64-byte-half swizzle of the
    // 128-byte records resolved for the four (bit 1, bit 0) classes of the compile-time record delta D
    int xbase[2][2];
    {
        const int L = 8 * (g >> 1) + q;
        const int cb = 64 * wci + 32 * (g & 1) + 8 * pp;
#pragma unroll
        for (int b1 = 0; b1 < 2; ++b1)
#pragma unroll
            for (int b0 = 0; b0 < 2; ++b0) {
                if constexpr (WCI == 2) {
                    const int bit1 = ((L >> 1) & 1) ^ b1 ^ (b0 & L & 1);   // bit 1 of (L + D)
                    xbase[b1][b0] = L * 128 + ((wci ^ bit1) << 6) + (cb & 63);
                } else {
                    xbase[b1][b0] = L * 64 + cb;
                }
            }
    }
    cv_f32x16 acc[RBW][KS];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][tx][i] = 0.f;

    const long long HW = (long long)P.H * P.W;
    const int t0 = (int)((long long)split * P.ntiles / P.ksplit), t1 = (int)((long long)(split + 1) * P.ntiles / P.ksplit);
    const int tpi = P.tiles_x * P.tiles_y;
    for (int tile = t0; tile < t1; ++tile) {
        const int n = tile / tpi, trem = tile - n * tpi;
        const int ty0 = (trem / P.tiles_x) * 2, tx0 = (trem % P.tiles_x) * CV_TW;
        const T* xn = x + (long long)n * P.Cin * HW;
        const T* dyn = dy + (long long)n * P.Cout * HW;
        __syncthreads();   // previous tile's readers are done
        // ---- X window of kernel row ty: task = (8-channel group, row, 8-column run)
        constexpr int NXT = (CIB / 8) * 2 * 10;
        if (tid < NXT) {
            const int run = tid % 10, r = (tid / 10) & 1, cg = tid / 20;
            const int gy = ty0 + r + ty - PAD, gx = tx0 - 8 + 8 * run;
            const bool inb = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            const int cbase = cislab * CIB + 8 * cg;
            // branch-free: a run outside the tensor reads the image's first run instead and is masked to zero afterwards (a
            // load under a lane-dependent condition ends its basic block with s_waitcnt vmcnt(0): the 8 requests of a task
            // were 8 serial round trips, as were the 4 of a dY thread below -- 12 per tile, 32 % of the matrix rate)
            uint4 v[8];
            uint32_t okm[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = inb && cbase + i < P.Cin;
                okm[i] = ok ? 0xffffffffu : 0u;
                v[i] = *reinterpret_cast<const uint4*>(xn + (ok ? ((long long)(cbase + i) * P.H + gy) * P.W + gx : 0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[i].x &= okm[i]; v[i].y &= okm[i]; v[i].z &= okm[i]; v[i].w &= okm[i];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t lo[4], hi[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t a = k == 0 ? v[2 * m].x : k == 1 ? v[2 * m].y : k == 2 ? v[2 * m].z : v[2 * m].w;
                    const uint32_t b = k == 0 ? v[2 * m + 1].x : k == 1 ? v[2 * m + 1].y : k == 2 ? v[2 * m + 1].z
                                                                                                  : v[2 * m + 1].w;
                    lo[m] = __builtin_amdgcn_perm(b, a, 0x05040100u);
                    hi[m] = __builtin_amdgcn_perm(b, a, 0x07060302u);
                }
                const int R = r * XW + 8 * run + 2 * k;
                *reinterpret_cast<uint4*>(Xt + cvw_xoff<WCI>(R, cg)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
                *reinterpret_cast<uint4*>(Xt + cvw_xoff<WCI>(R + 1, cg)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            }
        }
        // ---- dY tile: [co][2 rows x 64 px], 16-byte chunks
        constexpr int NYT = ROWS * 16;
#pragma unroll
        for (int i0 = 0; i0 < NYT; i0 += 4 * 256) {
            uint4 v[4];
            uint32_t okm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = i0 + j * 256 + tid;
                const int chunk = idx & 15, row = idx >> 4;
                const int co = coslab * ROWS + row, gy = ty0 + (chunk >> 3), gx = tx0 + 8 * (chunk & 7);
                const bool ok = idx < NYT && co < P.Cout && gy < P.H && gx < P.W;
                okm[j] = ok ? 0xffffffffu : 0u;
                v[j] = *reinterpret_cast<const uint4*>(dyn + (ok ? ((long long)co * P.H + gy) * P.W + gx : 0));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = i0 + j * 256 + tid;
                const int chunk = idx & 15, row = idx >> 4;
                v[j].x &= okm[j]; v[j].y &= okm[j]; v[j].z &= okm[j]; v[j].w &= okm[j];
                if (idx < NYT) *reinterpret_cast<uint4*>(Yt + row * 256 + ((chunk ^ (row & 15)) << 4)) = v[j];
            }
        }
        __syncthreads();
        // ---- 8 k-steps of 16 pixels; this wave takes every WK-th
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if ((s % WK) != wk) continue;
            cv_s16x8 A[RBW];
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const int row = 32 * (wco * RBW + rb) + c;
                A[rb] = *reinterpret_cast<const cv_s16x8*>(Yt + row * 256 + (((2 * s + h) ^ (row & 15)) << 4));
            }
            // k row of this lane's transposing read: pixel 16s + 8*(g>>1) + q (+4); window record = L + D with the
            // lane part L = 8*(g>>1) + q and the compile-time D = row*XW + 16*(s&3) + 8 - PAD + tx (+4)
#pragma unroll
            for (int tx = 0; tx < KS; ++tx) {
                const int Da = (s >> 2) * XW + 16 * (s & 3) + 8 - PAD + tx, Db = Da + 4;
                const cv_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (cv_lds_s16x4*)(Xt + xbase[(Da >> 1) & 1][Da & 1] + Da * REC));
                const cv_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (cv_lds_s16x4*)(Xt + xbase[(Db >> 1) & 1][Db & 1] + Db * REC));
                const cv_s16x8 B = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) acc[rb][tx] = CvMma<T>::run(A[rb], B, acc[rb][tx]);
            }
        }
    }
    // ---- k-step phases of one (wco, wci) add up inside the block, one tx slice (RBW*4 KB per wave) at a time
    const int wq = wave % (WCO * WCI);
    if constexpr (WK > 1) {
        static_assert(WCO * WCI * RBW * 4096 <= ROWS * 256, "Yt holds one tx slice per (wco, wci)");
        float* red = reinterpret_cast<float*>(Yt) + wq * (RBW * 16 * 64) + lane;
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
#pragma unroll
            for (int ph = 1; ph < WK; ++ph) {
                __syncthreads();
                if (wk == ph) {
#pragma unroll
                    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) red[(rb * 16 + reg) * 64] = acc[rb][tx][reg];
                }
                __syncthreads();
                if (wk == 0) {
#pragma unroll
                    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) acc[rb][tx][reg] += red[(rb * 16 + reg) * 64];
                }
            }
        }
        if (wk != 0) return;
    }
    // ---- slab: [split][ty][z][wq][rb][tx][reg][lane]
    float* dst = part + ((((long long)split * KS + ty) * P.nz + z) * (WCO * WCI) + wq) * (RBW * KS * 16 * 64) + lane;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) dst[((rb * KS + tx) * 16 + reg) * 64] = acc[rb][tx][reg];
}

// one thread per (ty, z, wco, wci, rb, tx, reg, lane): sums over the splits in order, scatters into dW
template <int KS, int RBW, int WCO, int WCI>
__global__ void __launch_bounds__(256) conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                CvWgParams P, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    constexpr int PER_WAVE = RBW * KS * 16 * 64;
    long long t = idx;
    const int lane = (int)(t & 63);
    t >>= 6;
    const int reg = (int)(t & 15);
    t >>= 4;
    const int tx = (int)(t % KS);
    t /= KS;
    const int rb = (int)(t % RBW);
    t /= RBW;
    const int wq = (int)(t % (WCO * WCI));   // wave index without the k-phase
    t /= (WCO * WCI);
    const int z = (int)(t % P.nz);
    const int ty = (int)(t / P.nz);
    const int wco = wq % WCO, wci = wq / WCO;
    const int coslab = z / P.ncislab, cislab = z % P.ncislab;
    const int co = coslab * (32 * RBW * WCO) + 32 * (wco * RBW + rb) + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int ci = cislab * (32 * WCI) + 32 * wci + (lane & 31);
    if (co >= P.Cout || ci >= P.Cin) return;
    const long long inner = ((long long)(rb * KS + tx) * 16 + reg) * 64 + lane;
    float sum = 0.f;
    const long long sstride = (long long)KS * P.nz * (WCO * WCI) * PER_WAVE;
    const float* src = part + (((long long)ty * P.nz + z) * (WCO * WCI) + wq) * PER_WAVE + inner;
    for (int split = 0; split < P.ksplit; ++split) sum += src[split * sstride];
    dw[(((long long)co * P.Cin + ci) * KS + ty) * KS + tx] = sum;
}

struct CvWgPlan {
    int cfg;   // WCI = 2 (64-channel slabs): 0: RBW2 WCO2 WK1, 1: RBW2 WCO1 WK2, 2: RBW1 WCO1 WK2;
               // WCI = 1 (<= 32 input channels per slab): 3: RBW2 WCO2 WK2 (as two k phases), 4: RBW2 WCO1 WK4, 5: RBW1 WCO1 WK4
    int rows, cib, rbw;
    CvWgParams P;
    size_t part_bytes;
    double note_bytes, note_flops;   // algorithmic bytes (16-bit activations) / flops of the launch (profile table)
};

static CvWgPlan cv_wg_plan(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K) {
    CvWgPlan p{};
    p.note_bytes = 2.0 * (double)N * (double)H * (double)W * (double)(Cin + Cout);
    p.note_flops = 2.0 * (double)N * (double)H * (double)W * (double)Cin * (double)Cout * K * K;
    const bool wide = Cin > 32;
    const int rcls = Cout > 64 ? 0 : (Cout > 32 ? 1 : 2);
    p.cfg = (wide ? 0 : 3) + rcls;
    p.rows = rcls == 0 ? 128 : (rcls == 1 ? 64 : 32);
    p.rbw = rcls == 2 ? 1 : 2;
    p.cib = wide ? 64 : 32;
    CvWgParams& P = p.P;
    P.Cin = (int)Cin; P.Cout = (int)Cout; P.H = (int)H; P.W = (int)W;
    P.tiles_x = (int)cdiv(W, CV_TW); P.tiles_y = (int)cdiv(H, 2);
    P.ntiles = (int)(N * P.tiles_x * P.tiles_y);
    P.ncislab = (int)cdiv(Cin, p.cib);
    P.nz = (int)cdiv(Cout, p.rows) * P.ncislab;
    // two blocks per CU, but at least 8 pixel tiles per block (every split costs a 40 KB slab per wave)
    int ks = 512 / (K * P.nz);
    ks = ks > P.ntiles / 8 ? P.ntiles / 8 : ks;
    if (ks > 8) ks -= ks % 8;     // whole rounds of the 8 XCDs
    P.ksplit = ks < 1 ? 1 : ks;
    const int nwq = wide ? (rcls == 0 ? 4 : 2) : (rcls == 0 ? 2 : 1);   // (wco, wci) pairs per block
    p.part_bytes = (size_t)P.ksplit * K * P.nz * nwq * (p.rbw * K * 16 * 64) * sizeof(float);
    return p;
}

template <typename T>
static int launch_conv2d_wgrad(const char* name, const void* dy, const void* x, float* dw, int K, const CvWgPlan& p, void* ws,
                               hipStream_t st) {
    const CvWgParams& P = p.P;
    dim3 grid((unsigned)(cdiv(P.ksplit, 8) * 8 * K * P.nz));
    prof_note(p.note_bytes, p.note_flops);
#define OFASR_WG(KS, RBW, WCO, WCI, WK)                                                                              \
    {                                                                                                                \
        OFASR_LAUNCH((conv_wgrad_kernel<T, KS, RBW, WCO, WCI, WK>), grid, dim3(256), 0, st, (const T*)dy,      \
                           (const T*)x, (float*)ws, P);                                                              \
        int rc = check_launch(name);                                                                                 \
        if (rc) return rc;                                                                                           \
        const long long tot = (long long)KS * P.nz * (WCO * WCI) * RBW * KS * 16 * 64;                               \
        OFASR_LAUNCH((conv_wgrad_reduce_kernel<KS, RBW, WCO, WCI>), dim3((unsigned)cdiv(tot, 256)),            \
                           dim3(256), 0, st, (const float*)ws, dw, P, tot);                                          \
    }
#define OFASR_WGK(KS)                                                                                                \
    switch (p.cfg) {                                                                                                 \
        case 0: OFASR_WG(KS, 2, 2, 2, 1) break;                                                                      \
        case 1: OFASR_WG(KS, 2, 1, 2, 2) break;                                                                      \
        case 2: OFASR_WG(KS, 1, 1, 2, 2) break;                                                                      \
        case 3: OFASR_WG(KS, 2, 2, 1, 2) break;                                                                      \
        case 4: OFASR_WG(KS, 2, 1, 1, 4) break;                                                                      \
        default: OFASR_WG(KS, 1, 1, 1, 4) break;                                                                     \
    }
    if (K == 5) { OFASR_WGK(5) } else { OFASR_WGK(3) }
#undef OFASR_WGK
#undef OFASR_WG
    return check_launch(name);
}

static int conv_stat_units(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K) {
    if (conv_thin_out_supported(Cout, Cin, K, W, OFASR_BF16, nullptr, nullptr))   // the head: one partial per workgroup
        return conv_thin_out_units(N, Cout, Cin, H, W, OFASR_BF16);
    if (conv_thin_in_supported(Cin, Cout, K, W, OFASR_BF16, nullptr, nullptr))    // the stem
        return conv_thin_in_units(N, Cin, Cout, H, W, OFASR_BF16);
    const CvPlan p = cv_plan(Cin, Cout, K, 0, N, H, W);
    const int wp = p.th / 2;     // waves along the pixel rows of a tile: 2 rows each
    return (int)(N * cdiv(W, CV_TW) * cdiv(H, p.th) * wp);
}

static int conv2d_entry(const char* name, const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout,
                        int64_t H, int64_t W, int K, int dtype, int dgrad, void* ws, size_t ws_bytes, void* stream,
                        StatOut so = StatOut{nullptr, 0}) {
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
    {
        // a 3-channel result (the head's forward, the stem's input gradient): csrc/conv_thin.hip
        const int64_t Ct = dgrad ? Cin : Cout, Cw = dgrad ? Cout : Cin;
        if (conv_thin_out_supported(Ct, Cw, K, W, dtype, x, y)) {
            if (so.partial)
                OFASR_REQUIRE(!dgrad && so.P == conv_thin_out_units(N, Ct, Cw, H, W, dtype), OFASR_ERR_INVALID_ARG,
                              "%s: statistics unit count %d does not match the launch", name, so.P);
            return conv_thin_out(x, w, y, N, Ct, Cw, H, W, K, dtype, dgrad, so, stream);
        }
        // a 3-channel operand (the stem's forward, the head's input gradient)
        if (conv_thin_in_supported(Cw, Ct, K, W, dtype, x, y)) {
            if (so.partial)
                OFASR_REQUIRE(!dgrad && so.P == conv_thin_in_units(N, Cw, Ct, H, W, dtype), OFASR_ERR_INVALID_ARG,
                              "%s: statistics unit count %d does not match the launch", name, so.P);
            return conv_thin_in(x, w, y, N, Cw, Ct, H, W, K, dtype, dgrad, so, stream);
        }
    }
    if (so.partial)
        OFASR_REQUIRE(!dgrad && so.P == conv_stat_units(N, Cin, Cout, H, W, K), OFASR_ERR_INVALID_ARG,
                      "%s: statistics unit count %d does not match the launch", name, so.P);
    if (dtype == OFASR_BF16) return launch_conv2d<bf16_t>(name, x, w, y, N, Cin, Cout, H, W, K, dgrad, ws, st, so);
    return launch_conv2d<f16_t>(name, x, w, y, N, Cin, Cout, H, W, K, dgrad, ws, st, so);
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

// ---- inference: conv + eval-mode BN (+ ReLU6 | PixelShuffle(2)) as one kernel, operands prepared once per set of weights
static size_t cv_infer_bytes(int64_t Cin, int64_t Cout, int K) {
    return (cv_plan(Cin, Cout, K, 0).img_bytes + 255) / 256 * 256 + (size_t)(2 * Cout) * sizeof(float);
}
OFASR_EXPORT size_t ofasr_conv2d_infer_operand_bytes(int64_t Cin, int64_t Cout, int K) {
    if (Cin <= 0 || Cout <= 0 || !(K == 3 || K == 5)) return 0;
    return cv_infer_bytes(Cin, Cout, K);
}

OFASR_EXPORT int ofasr_conv2d_infer_prepare(const float* w, const float* gamma, const float* beta, const float* running_mean,
                                            const float* running_var, double eps, int64_t Cin, int64_t Cout, int K, int dtype,
                                            void* operands, size_t operand_bytes, void* stream) {
    const char* name = "ofasr_conv2d_infer_prepare";
    OFASR_REQUIRE(w && operands, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(Cin > 0 && Cout > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit activations only", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(gamma == nullptr || (beta && running_mean && running_var), OFASR_ERR_INVALID_ARG, "%s: incomplete BN", name);
    OFASR_REQUIRE(operand_bytes >= cv_infer_bytes(Cin, Cout, K), OFASR_ERR_WORKSPACE, "%s: operand buffer %zu B < required %zu B",
                  name, operand_bytes, cv_infer_bytes(Cin, Cout, K));
    float* ss = reinterpret_cast<float*>((char*)operands + (cv_plan(Cin, Cout, K, 0).img_bytes + 255) / 256 * 256);
    const CvBn bn{ss, gamma, beta, running_mean, running_var, (float)eps};
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_BF16) return launch_conv_prep<bf16_t>(name, w, Cin, Cout, K, 0, operands, bn, st);
    return launch_conv_prep<f16_t>(name, w, Cin, Cout, K, 0, operands, bn, st);
}

OFASR_EXPORT int ofasr_conv2d_infer_run(const void* x, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                        int K, int dtype, int act, const void* operands, size_t operand_bytes, void* stream) {
    const char* name = "ofasr_conv2d_infer_run";
    OFASR_REQUIRE(x && y && operands, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit activations only", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(act >= 0 && act <= 2, OFASR_ERR_INVALID_ARG, "%s: act %d not in {0 none, 1 relu6, 2 pixel shuffle}", name, act);
    OFASR_REQUIRE(act != 2 || Cout % 4 == 0, OFASR_ERR_INVALID_ARG, "%s: PixelShuffle(2) needs Cout %% 4 == 0", name);
    OFASR_REQUIRE(W % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: needs W %% 8 == 0 and 16-byte aligned tensors", name);
    OFASR_REQUIRE(N <= 65535 && H * W <= (1LL << 29), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    OFASR_REQUIRE(operand_bytes >= cv_infer_bytes(Cin, Cout, K), OFASR_ERR_WORKSPACE, "%s: operand buffer %zu B < required %zu B",
                  name, operand_bytes, cv_infer_bytes(Cin, Cout, K));
    const float* ss = reinterpret_cast<const float*>((const char*)operands + (cv_plan(Cin, Cout, K, 0).img_bytes + 255) / 256 * 256);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_BF16) return launch_conv_run<bf16_t>(name, x, operands, y, N, Cin, Cout, H, W, K, 0, CvEpi{ss, act, nullptr, 0}, st);
    return launch_conv_run<f16_t>(name, x, operands, y, N, Cin, Cout, H, W, K, 0, CvEpi{ss, act, nullptr, 0}, st);
}

// forward that also leaves the BatchNorm statistics partials of its output: partial[Cout][units] (sum, sum of squares),
// units = ofasr_conv2d_stat_units(...); fold them with ofasr_bn_finalize_cp / ofasr_bn_fwd_cp
OFASR_EXPORT int ofasr_conv2d_stat_units(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !(K == 3 || K == 5)) return 0;
    return conv_stat_units(N, Cin, Cout, H, W, K);
}

OFASR_EXPORT int ofasr_conv2d_fwd_stat(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                       int64_t W, int K, int dtype, void* partial, int64_t units, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    OFASR_REQUIRE(partial != nullptr && units > 0 && units <= INT32_MAX, OFASR_ERR_INVALID_ARG,
                  "ofasr_conv2d_fwd_stat: null / empty statistics buffer");
    return conv2d_entry("ofasr_conv2d_fwd_stat", x, w, y, N, Cin, Cout, H, W, K, dtype, 0, workspace, workspace_bytes, stream,
                        StatOut{(float2*)partial, (int)units});
}

OFASR_EXPORT int ofasr_conv2d_dgrad(const void* dy, const float* w, void* dx, int64_t N, int64_t Cin, int64_t Cout,
                                    int64_t H, int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    return conv2d_entry("ofasr_conv2d_dgrad", dy, w, dx, N, Cin, Cout, H, W, K, dtype, 1, workspace, workspace_bytes,
                        stream);
}

OFASR_EXPORT size_t ofasr_conv2d_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || !(K == 3 || K == 5)) return 0;
    size_t need = cv_wg_plan(N, Cin, Cout, H, W, K).part_bytes;
    const int64_t Ct = Cin < Cout ? Cin : Cout, Cw = Cin < Cout ? Cout : Cin;
    if (conv_thin_wgrad_supported(Ct, Cw, K, H, W, OFASR_BF16, nullptr, nullptr)) {   // the head / stem: csrc/conv_thin.hip
        const size_t thin = conv_thin_wgrad_workspace(N, Ct, Cw, H, W, K, OFASR_BF16);
        need = thin > need ? thin : need;
    }
    return need;
}

OFASR_EXPORT int ofasr_conv2d_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout,
                                    int64_t H, int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    const char* name = "ofasr_conv2d_wgrad";
    OFASR_REQUIRE(dy && x && dw, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED,
                  "%s: 16-bit activations only (fp32 runs on the vendor library)", name);
    OFASR_REQUIRE(K == 3 || K == 5, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {3,5}", name, K);
    OFASR_REQUIRE(W % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: needs W %% 8 == 0 and 16-byte aligned tensors", name);
    OFASR_REQUIRE(N * H * W <= (1LL << 30), OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    {
        const int64_t Ct = Cin < Cout ? Cin : Cout, Cw = Cin < Cout ? Cout : Cin;
        if (conv_thin_wgrad_supported(Ct, Cw, K, H, W, dtype, dy, x))
            return conv_thin_wgrad(dy, x, dw, N, Cin, Cout, H, W, K, dtype, workspace, workspace_bytes, stream);
    }
    const CvWgPlan p = cv_wg_plan(N, Cin, Cout, H, W, K);
    OFASR_REQUIRE(workspace && workspace_bytes >= p.part_bytes, OFASR_ERR_WORKSPACE,
                  "%s: workspace %zu B < required %zu B", name, workspace_bytes, p.part_bytes);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_BF16) return launch_conv2d_wgrad<bf16_t>(name, dy, x, dw, K, p, workspace, st);
    return launch_conv2d_wgrad<f16_t>(name, dy, x, dw, K, p, workspace, st);
}
