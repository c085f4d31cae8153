// mbfused.hip -- the WHOLE MB block as ONE kernel for eval-mode / frozen BatchNorm (ofasr_mbconv_infer):
//     out = x + BN3(W2 . relu6(BN2(dw_k(relu6(BN1(W1 . x))))))
// DynamicMBConvLayer.forward + the identity shortcut (reference ofa/elastic_nn/modules/dynamic_layers.py:70-84,
// ofa/imagenet_codebase/networks/proxyless_nets.py:44-51) in the BN regime the reference validates in
// (sr_run_manager.py:323-393 net.eval(), eval_ofa_net_sr.py:187-220) -- and freezes its teacher BN in (:417-420).
//
// With running statistics every BN is an affine map, so it folds into its conv: W1f = s1.W1 (rows), b1;
// taps f.s2, b2; W2f = s3.W2, b3 (mb_fold_kernel; 16-bit operands, fp32 biases).  The block then reads x ONCE and writes
// out ONCE: 2 x 8.4 MB per call at N=16, 64x64 (25 MB with the shortcut read) instead of the 243 MB the un-fused eval path
// moves through y1 / y2 / y3 -- the mid tensor (384 channels) never leaves the CU.  This is the variant in which the
// 1x1 path is matrix-bound.
//
// One workgroup (8 waves) = one 16x16 output tile of one image, all channels:
//   prologue  the x window (16+2P rows x 24 columns, image columns w0-4 .. w0+19: 8-byte aligned quads) of all 64
//             channels -> LDS; every wave pulls the MFMA fragments of its <= 3 blocks of 32 window pixels into
//             registers (transposing reads); the LDS image is then dead and its space reused
//   per chunk of 32 mid channels (mid/32 chunks), software-pipelined over two barriers:
//     E  expand on the matrix cores: a1[32 px x 32 ch] = X^T[32 px x 64] . W1f^T[64 x 32] per pixel block, + b1, ReLU6,
//        positions outside the image forced to 0 (the depthwise conv pads the ACTIVATED tensor with zeros), 16-bit,
//        written as channel planes (4 adjacent pixels per lane: the accumulator rows) -> A1[buf]
//     D  depthwise k x k on the vector pipe from the planes: a lane owns 4 adjacent outputs of one row; two taps per
//        v_dot2c_f32_{bf16,f16} (fp32 accumulation of exact 16-bit products) against the pair table (f0 f1)(f2 f3)..;
//        outputs whose window starts on an odd column read the row shifted by one (v_alignbit), so one table serves
//        all; taps are wave-uniform scalar registers, the next channel's row requested a channel ahead; + b2, ReLU6,
//        16-bit -> A2
//     P  project on the matrix cores: out[32 px x 64] += a2^T[32 px x 32 ch] . W2f^T[32 x 64]  (accumulators resident)
//     waves 4-7 run D before E so that the two waves of a SIMD are on different pipes (matrix || vector)
//   epilogue  + b3 (+ x), staged through LDS so that the tile leaves as 16-byte row pieces.
//
// Bounds (N=16, 64x64, mid 384, bf16): HBM 25 MB -> 4.6 us; matrix work 2*(1.89*64*384 + 384*64)*65536*... = 9.3 GFLOP
// at k=7 (the window re-computes the expand 1.89x; 7.3 GFLOP at k=3) -> 3.7 us at the 2.5 PF peak; depthwise k=7
// 28 v_dot2c per output = 43 k wave-instructions per CU.  DESIGN.md section 3 has the measured numbers.
#include "ofasr_common.h"

#include <algorithm>
#include <atomic>

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 __attribute__((aligned(2))) u32x2_u;   // vector requests at any element boundary (ragged image widths)
typedef u32x4 __attribute__((aligned(2))) u32x4_u;

constexpr int MF_THREADS = 512;
constexpr int MF_MC = 32;         // mid channels per chunk
constexpr int MF_A2P = 288;       // pixel pitch of the a2 planes (576 B = 64 mod 256)
constexpr int MF_SP = 260;        // pixel pitch of the fp32 output stage

// The output tile is TH x TW = 256 pixels: 16x16 (least halo), 8x32 or 4x64 (whole 128-byte lines of a 64-wide image per
// row piece).  The window is TH+2P rows x TW+8 columns (image columns w0-4 .. w0+TW+3: 8-byte aligned quads).
template <int K, int TH_, int TW_> struct MfGeom {
    static_assert(TH_ * TW_ == 256 && TW_ % 8 == 0, "tile = 256 pixels, rows of 16-byte pieces");
    static constexpr int TH = TH_, TW = TW_;
    static constexpr int WC = TW + 8;                       // window columns
    static constexpr int QW = WC / 4;                       // quads per window row
    static constexpr int P = K / 2;
    static constexpr int HT = TH + 2 * P;                   // window rows
    static constexpr int NPIX = HT * WC;                    // window pixels (16x16: 528 / 480 / 432)
    static constexpr int NBLK = (NPIX + 31) / 32;           // 17 / 15 / 14
    static constexpr int NB_WAVE = (NBLK + 7) / 8;          // pixel blocks per wave (3 / 2 / 2)
    // pixel pitch of the x window planes: >= 32 NBLK and = 32 (mod 128), i.e. 64 bytes (mod 256): the four rows of a
    // transposing read then sit on disjoint banks
    static constexpr int XP = (NBLK * 32 - 32 + 127) / 128 * 128 + 32;
    static constexpr int A1P = NBLK * 32 + 4;               // plane pitch: = 4 (mod 8) -> 8-byte stores of 16 planes spread over banks
    static constexpr int NPAIR = (K + 1) / 2;               // tap pairs per kernel row and parity
    static constexpr int TAPS = K * NPAIR;                  // tap-pair dwords per channel: (f0 f1)(f2 f3)..(f_{K-1} 0) per kernel row
    static constexpr int TAPROW = (TAPS + 1 + 3) / 4 * 4;   // + the BN2 bias, padded: one scalar-load row per channel
};

template <typename T> struct Mf;
template <> struct Mf<bf16_t> {
    static __device__ __forceinline__ f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t tap, uint32_t v, float acc) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, tap), __builtin_bit_cast(bf16x2, v), acc, false);
    }
};
template <> struct Mf<f16_t> {
    static __device__ __forceinline__ f32x16 mma(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t tap, uint32_t v, float acc) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, tap), __builtin_bit_cast(f16x2, v), acc, false);
    }
};

// fragment of 8 k-values (rows kb .. kb+7 of a [k][pixel] LDS image with `pitch` bytes per row) of pixel column
// pos0 + (lane & 31): lane (r, h) gets k = 16 s + 8 h + 0..7 -- the A operand (row = pixel) and the B operand
// (column = pixel) of the 32x32x16 MFMA want exactly this.  Two transposing reads.
__device__ __forceinline__ s16x8 mf_frag(const char* img, int pitch, int pos0, int s, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kb = 16 * s + 8 * (g >> 1);
    const int colb = (pos0 + 16 * (g & 1) + 4 * pp) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (kb + q) * pitch + colb));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (kb + 4 + q) * pitch + colb));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ int mf_acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- BN folding: the per-call 16-bit operand images of the fused kernel --------------------------------------
struct MfFold {
    const float* w1; long long ldw1;
    const float* w2; long long ldw2;
    const float* f;                      // active depthwise filter [mid][K][K] (ofasr_ktransform_fwd)
    const float* gamma[3]; const float* beta[3]; const float* mean[3]; const float* var[3];
    float eps[3];
    int mid, K;
};

template <typename T>
__global__ void __launch_bounds__(256) mb_fold_kernel(MfFold p, T* __restrict__ w1f, float* __restrict__ b1,
                                                      uint32_t* __restrict__ taps, float* __restrict__ b2,
                                                      T* __restrict__ w2f, float* __restrict__ b3) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int mid = p.mid, K = p.K, NPAIR = (K + 1) / 2;
    auto scale = [&](int i, int c) { return p.gamma[i][c] * rsqrtf(p.var[i][c] + p.eps[i]); };
    // Weight images in MFMA-fragment order: the 16 bytes lane l of a wave needs for (chunk, k-step) sit at
    // [(chunk, s)][l], so a fragment load is one fully coalesced 1 KB request.
    //   w1f[((ci*4 + s)*64 + l)*8 + j] = s1[c] * W1[c][16 s + 8 (l>>5) + j],          c = 32 ci + (l & 31)
    //   w2f[(((ci*2 + ob)*2 + s)*64 + l)*8 + j] = s3[o] * W2[o][32 ci + 16 s + 8 (l>>5) + j],  o = 32 ob + (l & 31)
    for (int e = tid; e < mid * 64; e += nth) {
        const int j = e & 7, l = (e >> 3) & 63, s4 = (e >> 9) & 3, ci = e >> 11;
        const int c = 32 * ci + (l & 31), k = 16 * s4 + 8 * (l >> 5) + j;
        w1f[e] = from_float<T>(p.w1[(long long)c * p.ldw1 + k] * scale(0, c));
    }
    for (int e = tid; e < 64 * mid; e += nth) {
        const int j = e & 7, l = (e >> 3) & 63, s2 = (e >> 9) & 1, ob = (e >> 10) & 1, ci = e >> 11;
        const int o = 32 * ob + (l & 31), c = 32 * ci + 16 * s2 + 8 * (l >> 5) + j;
        w2f[e] = from_float<T>(p.w2[(long long)o * p.ldw2 + c] * scale(2, o));
    }
    const int TAPROW = (K * NPAIR + 1 + 3) / 4 * 4;
    for (int c = tid; c < mid; c += nth) {
        b1[c] = p.beta[0][c] - p.mean[0][c] * scale(0, c);
        b2[c] = p.beta[1][c] - p.mean[1][c] * scale(1, c);
        taps[(long long)c * TAPROW + K * NPAIR] = __float_as_uint(b2[c]);   // the row's last word: BN2 bias
    }
    for (int o = tid; o < 64; o += nth) b3[o] = p.beta[2][o] - p.mean[2][o] * scale(2, o);
    // tap pairs per (channel, kernel row): (f0 f1)(f2 f3)..(f_{K-1} 0); the low half multiplies the lower column
    for (int e = tid; e < mid * K * NPAIR; e += nth) {
        const int c = e / (K * NPAIR), r = e - c * (K * NPAIR);
        const int ky = r / NPAIR, m = r - ky * NPAIR;
        const float s = scale(1, c);
        const float* fr = p.f + ((long long)c * K + ky) * K;
        const float lo = fr[2 * m] * s;
        const float hi = (2 * m + 1 < K) ? fr[2 * m + 1] * s : 0.f;
        taps[(long long)c * TAPROW + r] = (uint32_t)from_float<T>(lo).v | ((uint32_t)from_float<T>(hi).v << 16);
    }
}

// ---- the fused block ---------------------------------------------------------------------------------------
template <typename T, int K, int TH, int TW>
__global__ void __launch_bounds__(MF_THREADS) mb_fused_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                              const T* __restrict__ w1f, const float* __restrict__ b1,
                                                              const uint32_t* __restrict__ taps,
                                                              const float* __restrict__ b2, const T* __restrict__ w2f,
                                                              const float* __restrict__ b3, int mid, int H, int W,
                                                              int tiles_x, int tiles_y, int residual, int nsplit,
                                                              float* __restrict__ part) {
    using G = MfGeom<K, TH, TW>;
    constexpr int P = G::P, HT = G::HT, NPIX = G::NPIX, NBLK = G::NBLK, NBW = G::NB_WAVE, A1P = G::A1P, NPAIR = G::NPAIR;
    constexpr int MF_WC = G::WC, MF_XP = G::XP, QW = G::QW;
    constexpr int A1_BYTES = MF_MC * A1P * 2, A2_BYTES = MF_MC * MF_A2P * 2;
    constexpr int X_BYTES = 64 * MF_XP * 2, S_BYTES = 64 * MF_SP * 4;
    constexpr int BODY = 2 * A1_BYTES + A2_BYTES;
    constexpr int LDS_BYTES = BODY > X_BYTES ? (BODY > S_BYTES ? BODY : S_BYTES) : (X_BYTES > S_BYTES ? X_BYTES : S_BYTES);
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    char* Xs = lds;                      // prologue only
    char* A1 = lds;                      // two buffers
    char* A2 = lds + 2 * A1_BYTES;
    float* St = reinterpret_cast<float*>(lds);   // epilogue only

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    // nsplit > 1 (launches with fewer tiles than CUs): nsplit adjacent workgroups share a tile, each takes a contiguous
    // range of the mid-channel chunks and leaves its partial projection in `part`; mb_fused_sum_kernel adds them up.
    // (Folding them in the workgroup that arrives last -- arrival counter + device-scope fences -- was measured at 118 us
    // per launch against 68 unsplit: every fence writes back and invalidates the XCD's whole L2 under the other tiles.)
    const int tile = nsplit > 1 ? blockIdx.x / nsplit : blockIdx.x;
    const int split = nsplit > 1 ? blockIdx.x - tile * nsplit : 0;
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int h0 = ty * TH, w0 = tx * TW;
    const long long plane = (long long)H * W;
    const T* xn = x + (long long)n * 64 * plane;

    // ---- prologue: x window -> LDS (zeros outside the image and beyond the window)
    {
        constexpr int QUADS = 64 * HT * QW;
        constexpr int NIT = (QUADS + MF_THREADS - 1) / MF_THREADS;
        if (W >= 4) {
            // Request ALL quads from clamped (always valid) addresses as straight-line code and fix them up afterwards --
            // a load under a lane-dependent branch would end its basic block with s_waitcnt vmcnt(0): 17 serial round
            // trips instead of one.  Rows of a ragged width start at any 2-byte boundary: the loads are declared
            // 2-byte aligned (global memory takes unaligned dwordx2 requests).  A quad cut by the right border is read
            // `sh` elements early and shifted down, which also zero-fills its tail.
            uint2 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e0 = tid + it * MF_THREADS;
                const int e = e0 < QUADS ? e0 : QUADS - 1;
                const int c = e / (HT * QW), rem = e - c * (HT * QW);
                const int hh = rem / QW, qd = rem - hh * QW;
                const int gh = h0 - P + hh, gw = w0 - 4 + 4 * qd;
                const int ghc = gh < 0 ? 0 : (gh >= H ? H - 1 : gh), gwc = gw < 0 ? 0 : (gw + 4 > W ? W - 4 : gw);
                const u32x2 q = *reinterpret_cast<const u32x2_u*>(xn + (long long)c * plane + (long long)ghc * W + gwc);
                v[it] = make_uint2(q.x, q.y);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * MF_THREADS;
                if (e < QUADS) {
                    const int c = e / (HT * QW), rem = e - c * (HT * QW);
                    const int hh = rem / QW, qd = rem - hh * QW;
                    const int gh = h0 - P + hh, gw = w0 - 4 + 4 * qd;
                    const bool ok = gh >= 0 && gh < H && gw >= 0 && gw < W;
                    const int sh = gw + 4 > W ? gw + 4 - W : 0;     // 1..3 on the one quad the right border cuts
                    const unsigned long long q = (((unsigned long long)v[it].y << 32) | v[it].x) >> (16 * sh);
                    *reinterpret_cast<uint2*>(Xs + c * (MF_XP * 2) + rem * 8) =
                        ok ? make_uint2((uint32_t)q, (uint32_t)(q >> 32)) : make_uint2(0u, 0u);
                }
            }
        } else {   // images narrower than one quad: element-wise guarded loads straight into the LDS image
            for (int e = tid; e < QUADS; e += MF_THREADS) {
                const int c = e / (HT * QW), rem = e - c * (HT * QW);
                const int hh = rem / QW, qd = rem - hh * QW;
                const int gh = h0 - P + hh, gw = w0 - 4 + 4 * qd;
                uint2 q = make_uint2(0u, 0u);
                if (gh >= 0 && gh < H) {
                    const T* src = xn + (long long)c * plane + (long long)gh * W + gw;
                    const uint32_t e0 = (gw >= 0 && gw < W) ? src[0].v : 0u, e1 = (gw + 1 >= 0 && gw + 1 < W) ? src[1].v : 0u;
                    const uint32_t e2 = (gw + 2 >= 0 && gw + 2 < W) ? src[2].v : 0u, e3 = (gw + 3 >= 0 && gw + 3 < W) ? src[3].v : 0u;
                    q = make_uint2(e0 | (e1 << 16), e2 | (e3 << 16));
                }
                *reinterpret_cast<uint2*>(Xs + c * (MF_XP * 2) + rem * 8) = q;
            }
        }
        // the tail of every plane (window pixels NPIX .. XP): read by the last pixel block's fragments
        for (int e = tid; e < 64 * ((MF_XP - NPIX) / 4); e += MF_THREADS) {
            const int c = e / ((MF_XP - NPIX) / 4), qd = e - c * ((MF_XP - NPIX) / 4);
            *reinterpret_cast<uint2*>(Xs + c * (MF_XP * 2) + (NPIX + 4 * qd) * 2) = make_uint2(0u, 0u);
        }
    }
    __syncthreads();
    // X fragments of this wave's pixel blocks (block pb = wave + 8 j) and, per block, the 16-bit AND masks of its 8 packed
    // output words (positions outside the image / beyond the window are forced to zero after the activation).  Masks are
    // VGPR data on purpose: as predicates the compiler keeps 16 lane masks per block in SGPR pairs and spills them.
    s16x8 xf[NBW][4];
    uint32_t mk[NBW][8];
#pragma unroll
    for (int j = 0; j < NBW; ++j) {
        const int pb = wave + 8 * j;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            xf[j][s] = pb < NBLK ? mf_frag(Xs, MF_XP * 2, 32 * (pb < NBLK ? pb : 0), s, lane) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int wd = 0; wd < 8; ++wd) {
            uint32_t m = 0u;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int reg = 2 * wd + e;
                const int px = 32 * pb + mf_acc_row(reg, h);
                const int hh = px / MF_WC, ww = px - hh * MF_WC;
                const int gh = h0 - P + hh, gw = w0 - 4 + ww;
                if (px < NPIX && gh >= 0 && gh < H && gw >= 0 && gw < W) m |= 0xffffu << (16 * e);
            }
            mk[j][wd] = m;
        }
    }
    __syncthreads();   // Xs is dead from here on: A1 / A2 take its place

    f32x16 oacc[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[ob][i] = 0.f;

    const int nchunk = mid / MF_MC;
    const int q4 = lane % (TW / 4), row16 = lane / (TW / 4);   // depthwise: outputs (row16, 4 q4 .. 4 q4 + 3)
    const s16x8* w1q = reinterpret_cast<const s16x8*>(w1f) + lane;   // fragment-ordered images: [(chunk, s)][lane]
    const s16x8* w2q = reinterpret_cast<const s16x8*>(w2f) + lane;

    // operands: the expand fragments of chunk i+1 are requested at the top of iteration i (a whole chunk ahead); the
    // project fragments of chunk i-1 at the top of iteration i, i.e. a whole expand/depthwise phase before their use
    struct W1 {
        s16x8 w[4];
        float bias;
    };
    struct W2 {
        s16x8 w[2][2];
    };
    auto load_w1 = [&](int ci, W1& cw) {
#pragma unroll
        for (int s = 0; s < 4; ++s) cw.w[s] = w1q[(ci * 4 + s) * 64];
        cw.bias = b1[ci * MF_MC + r32];
    };
    auto load_w2 = [&](int ci, W2& cw) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int s = 0; s < 2; ++s) cw.w[ob][s] = w2q[((ci * 2 + ob) * 2 + s) * 64];
    };

    auto expand = [&](int ci, const W1& cw) {
        char* dst = A1 + (ci & 1) * A1_BYTES;
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int pb = wave + 8 * j;
            if (pb < NBLK) {   // wave-uniform
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = cw.bias;   // the folded BN1 shift rides in the accumulator
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = Mf<T>::mma(xf[j][s], cw.w[s], acc);
                char* pl = dst + r32 * (A1P * 2) + (32 * pb + 4 * h) * 2;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_fmed3f(acc[4 * g + i], 0.f, 6.f);
                    *reinterpret_cast<uint2*>(pl + 16 * g) =
                        make_uint2(pack2<T>(v[0], v[1]) & mk[j][2 * g], pack2<T>(v[2], v[3]) & mk[j][2 * g + 1]);
                }
            }
        }
    };

    auto depthwise = [&](int ci) {
        const int c0 = ci * MF_MC;
        const char* src = A1 + (ci & 1) * A1_BYTES;
        constexpr int NCH = MF_MC / 8;          // channels per wave and chunk
        constexpr int NTW = G::TAPS + 1;
        // The taps are wave-uniform and live in scalar registers; the row of channel i+1 is requested before channel i
        // is computed (two sets fit: 29 words at K = 7), so the scalar-load latency -- the phase runs at two waves per
        // SIMD and was bound by exactly this wait -- is covered by the previous channel's products.
        uint32_t tc[NTW], tn[NTW];
        {
            const uint32_t* tp = taps + (long long)(c0 + wave * NCH) * G::TAPROW;
#pragma unroll
            for (int q = 0; q < NTW; ++q) tc[q] = tp[q];
        }
        // window rows of the first PF kernel rows of a channel are requested one channel ahead as well (before the
        // previous channel's a2 store: both live in the one LDS array, so the compiler will not hoist them itself)
        constexpr int PF = K == 7 ? 3 : K;
        uint2 rw[K][3], rn[PF][3];
        auto load_rows = [&](int cc, int k0, int k1, uint2 (*dst)[3]) {
            const char* pl = src + cc * (A1P * 2) + (row16 * MF_WC + 4 * q4) * 2;
#pragma unroll
            for (int ky = k0; ky < k1; ++ky) {
                dst[ky - k0][0] = *reinterpret_cast<const uint2*>(pl + ky * (MF_WC * 2));
                dst[ky - k0][1] = *reinterpret_cast<const uint2*>(pl + ky * (MF_WC * 2) + 8);
                dst[ky - k0][2] = *reinterpret_cast<const uint2*>(pl + ky * (MF_WC * 2) + 16);
            }
        };
        load_rows(wave * NCH, 0, PF, rn);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cc = wave * NCH + i;
#pragma unroll
            for (int ky = 0; ky < PF; ++ky)
#pragma unroll
                for (int q = 0; q < 3; ++q) rw[ky][q] = rn[ky][q];
            if constexpr (PF < K) load_rows(cc, PF, K, rw + PF);
            if (i + 1 < NCH) {
                const uint32_t* tp = taps + (long long)(c0 + cc + 1) * G::TAPROW;
#pragma unroll
                for (int q = 0; q < NTW; ++q) tn[q] = tp[q];
                load_rows(cc + 1, 0, PF, rn);
            }
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const uint32_t d[6] = {rw[ky][0].x, rw[ky][0].y, rw[ky][1].x, rw[ky][1].y, rw[ky][2].x, rw[ky][2].y};   // window columns 4 q4 .. 4 q4 + 11
                // the same columns shifted by one: ds[m] = columns (2m+1, 2m+2), for the outputs whose window starts on an
                // odd column -- every output then uses the ONE pair table (f0 f1)(f2 f3)..
                uint32_t ds[5];
#pragma unroll
                for (int m = 0; m < 5; ++m) ds[m] = __builtin_amdgcn_alignbit(d[m + 1], d[m], 16);
#pragma unroll
                for (int m = 0; m < NPAIR; ++m)          // pair index outside: consecutive v_dot2c feed different accumulators
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int t = j + 4 - P;        // first window column of output j, relative to the lane's 12
                        const int base = t >> 1;
                        o[j] = Mf<T>::dot2(tc[ky * NPAIR + m], (t & 1) ? ds[base + m] : d[base + m], o[j]);
                    }
            }
            const float bias = __uint_as_float(tc[G::TAPS]);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_fmed3f(o[j] + bias, 0.f, 6.f);
            *reinterpret_cast<uint2*>(A2 + cc * (MF_A2P * 2) + lane * 8) =   // pixel row16 * TW + 4 q4 = 4 lane
                make_uint2(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]));
#pragma unroll
            for (int q = 0; q < NTW; ++q) tc[q] = tn[q];
        }
    };

    auto project = [&](const W2& cw) {
        s16x8 af[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) af[s] = mf_frag(A2, MF_A2P * 2, 32 * wave, s, lane);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int s = 0; s < 2; ++s) oacc[ob] = Mf<T>::mma(af[s], cw.w[ob][s], oacc[ob]);
    };

    // ---- chunk pipeline: [E(i) || D(i-1)] barrier [P(i-1)] barrier
    W1 cur, nxt;
    W2 pw;
    const int c_lo = nsplit > 1 ? split * nchunk / nsplit : 0;
    const int c_hi = nsplit > 1 ? (split + 1) * nchunk / nsplit : nchunk;
    load_w1(c_lo, cur);
    nxt = cur;
    for (int i = c_lo; i <= c_hi; ++i) {
        if (i + 1 < c_hi) load_w1(i + 1, nxt);
        if (i > c_lo) load_w2(i - 1, pw);
        if (wave < 4) {
            if (i < c_hi) expand(i, cur);
            if (i > c_lo) depthwise(i - 1);
        } else {
            if (i > c_lo) depthwise(i - 1);
            if (i < c_hi) expand(i, cur);
        }
        __syncthreads();
        if (i > c_lo) project(pw);
        __syncthreads();
        cur = nxt;
    }

    // ---- epilogue: + b3, stage fp32 [64][SP], then 16-byte row pieces (+ shortcut)
    if (nsplit > 1) {   // partial projection -> part[tile][split][64][256 pixels in tile order]
        float* mine = part + ((long long)tile * nsplit + split) * (64 * 256);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(mine + (32 * ob + r32) * 256 + 32 * wave + 8 * g + 4 * h) =
                    make_float4(oacc[ob][4 * g], oacc[ob][4 * g + 1], oacc[ob][4 * g + 2], oacc[ob][4 * g + 3]);
        return;
    } else {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const int o = 32 * ob + r32;
            const float bias = b3[o];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(St + o * MF_SP + 32 * wave + 8 * g + 4 * h) =
                    make_float4(oacc[ob][4 * g] + bias, oacc[ob][4 * g + 1] + bias, oacc[ob][4 * g + 2] + bias,
                                oacc[ob][4 * g + 3] + bias);
        }
    }
    __syncthreads();
    T* on = out + (long long)n * 64 * plane;
    constexpr int NPC = 64 * 16 * 2 / MF_THREADS;   // 16-byte row pieces per thread
    if (W >= 8) {
        // straight-line code: the shortcut pieces are requested together from clamped addresses (2-byte aligned vector
        // requests, as in the prologue); pieces wholly inside the image leave as one 16-byte store, the piece the right
        // border cuts element by element
        uint4 xr[NPC];
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int e = tid + it * MF_THREADS;
            const int half = e % (TW / 8), rr = (e / (TW / 8)) % TH, o = e >> 5;
            const int gh = h0 + rr, gw = w0 + 8 * half;
            const int ghc = gh < H ? gh : H - 1, gwc = gw + 8 <= W ? gw : W - 8;
            xr[it] = make_uint4(0u, 0u, 0u, 0u);
            if (residual) {
                const u32x4 q = *reinterpret_cast<const u32x4_u*>(xn + (long long)o * plane + (long long)ghc * W + gwc);
                xr[it] = make_uint4(q.x, q.y, q.z, q.w);
            }
        }
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int e = tid + it * MF_THREADS;
            const int half = e % (TW / 8), rr = (e / (TW / 8)) % TH, o = e >> 5;
            const int gh = h0 + rr, gw = w0 + 8 * half;
            const float* sp = St + o * MF_SP + rr * TW + 8 * half;
            const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
            float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const long long off = (long long)o * plane + (long long)gh * W + gw;
            if (gh < H && gw + 8 <= W) {
                const uint32_t xw[4] = {xr[it].x, xr[it].y, xr[it].z, xr[it].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    T lo, hi;
                    lo.v = (uint16_t)(xw[i] & 0xffffu);
                    hi.v = (uint16_t)(xw[i] >> 16);
                    v[2 * i] += to_float(lo);        // +0 when there is no shortcut (xr = 0 bits = +0.0 in both formats)
                    v[2 * i + 1] += to_float(hi);
                }
                *reinterpret_cast<u32x4_u*>(on + off) =
                    u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
            } else if (gh < H && gw < W) {
                for (int i = 0; i < W - gw; ++i) on[off + i] = from_float<T>(v[i] + (residual ? to_float(xn[off + i]) : 0.f));
            }
        }
    } else {   // images narrower than one piece: element-wise
        for (int e = tid; e < 64 * 16 * 2; e += MF_THREADS) {
            const int half = e % (TW / 8), rr = (e / (TW / 8)) % TH, o = e >> 5;
            const int gh = h0 + rr, gw = w0 + 8 * half;
            if (gh >= H || gw >= W) continue;
            const float* sp = St + o * MF_SP + rr * TW + 8 * half;
            const long long off = (long long)o * plane + (long long)gh * W + gw;
            for (int i = 0; i < 8 && gw + i < W; ++i) {
                const float xv = residual ? to_float(xn[off + i]) : 0.f;
                on[off + i] = from_float<T>(sp[i] + xv);
            }
        }
    }
}

// out = sum of the nsplit partial projections (fixed order) + b3 (+ x): one thread per 8-pixel row piece of a tile
template <typename T, int TH, int TW>
__global__ void __launch_bounds__(256) mb_fused_sum_kernel(const float* __restrict__ part, const T* __restrict__ x,
                                                           T* __restrict__ out, const float* __restrict__ b3, int H, int W,
                                                           int tiles_x, int tiles_y, int residual, int nsplit) {
    const int tile = blockIdx.x >> 3;                           // 2048 pieces per tile, 8 workgroups of 256
    const int e = ((blockIdx.x & 7) << 8) + threadIdx.x;
    const int half = e % (TW / 8), rr = (e / (TW / 8)) % TH, o = e >> 5;
    int b = tile;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int gh = ty * TH + rr, gw = tx * TW + 8 * half;
    const float* src = part + (long long)tile * nsplit * (64 * 256) + o * 256 + rr * TW + 8 * half;
    const float bias = b3[o];
    float v[8] = {bias, bias, bias, bias, bias, bias, bias, bias};
    for (int sp0 = 0; sp0 < nsplit; sp0 += 4) {                 // four partials requested together per round trip
        float4 q[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sp = sp0 + j < nsplit ? sp0 + j : nsplit - 1;
            q[j][0] = *reinterpret_cast<const float4*>(src + (long long)sp * (64 * 256));
            q[j][1] = *reinterpret_cast<const float4*>(src + (long long)sp * (64 * 256) + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (sp0 + j < nsplit) {                             // uniform
                v[0] += q[j][0].x; v[1] += q[j][0].y; v[2] += q[j][0].z; v[3] += q[j][0].w;
                v[4] += q[j][1].x; v[5] += q[j][1].y; v[6] += q[j][1].z; v[7] += q[j][1].w;
            }
    }
    if (gh >= H || gw >= W) return;
    const long long off = ((long long)n * 64 + o) * H * W + (long long)gh * W + gw;
    if (gw + 8 <= W) {
        if (residual) {
            const u32x4 q = *reinterpret_cast<const u32x4_u*>(x + off);
            float lo, hi;
            unpack2<T>(q.x, lo, hi); v[0] += lo; v[1] += hi;
            unpack2<T>(q.y, lo, hi); v[2] += lo; v[3] += hi;
            unpack2<T>(q.z, lo, hi); v[4] += lo; v[5] += hi;
            unpack2<T>(q.w, lo, hi); v[6] += lo; v[7] += hi;
        }
        *reinterpret_cast<u32x4_u*>(out + off) =
            u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
    } else {
        for (int i = 0; i < W - gw; ++i) out[off + i] = from_float<T>(v[i] + (residual ? to_float(x[off + i]) : 0.f));
    }
}

static size_t mf_align(size_t v) { return (v + 255) / 256 * 256; }
struct MfPlan {
    int tw, th, tiles_x, tiles_y, nsplit;
    long long tiles;
};
// Launches with fewer tiles than CUs split every tile's mid-channel chunks over `nsplit` workgroups: a workgroup's time
// is nchunk x (a latency-bound chunk), so a launch that cannot fill the chip is shortened by spreading the chunks.
// OFASR_MBFUSED_SPLIT=0 disables it.
constexpr int MF_CUS = 256;
static std::atomic<int> g_mf_split{[] { const char* e = getenv("OFASR_MBFUSED_SPLIT"); return (e && e[0] == '0') ? 0 : 1; }()};
static MfPlan mf_plan(const ofasr_mbconv_desc* d, int tw) {
    const bool split_ok = g_mf_split.load(std::memory_order_relaxed) != 0;
    MfPlan pl;
    pl.tw = tw;
    pl.th = 256 / tw;
    pl.tiles_x = (int)cdiv(d->W, tw);
    pl.tiles_y = (int)cdiv(d->H, pl.th);
    pl.tiles = (long long)d->N * pl.tiles_x * pl.tiles_y;
    const long long nchunk = d->mid / MF_MC;
    long long ns = split_ok && pl.tiles > 0 ? MF_CUS / pl.tiles : 1;
    if (ns > nchunk) ns = nchunk;
    pl.nsplit = ns < 2 ? 1 : (int)ns;
    return pl;
}
struct MfWs {
    size_t f, w1f, b1, taps, b2, w2f, b3, part, total;
};
static MfWs mf_ws(const ofasr_mbconv_desc* d) {
    const int64_t mid = d->mid;
    const int K = d->K;
    size_t n_part = 0;     // the largest of the three tile shapes (the shape is chosen at launch time)
    for (int tw = 16; tw <= 64; tw *= 2) {
        const MfPlan pl = mf_plan(d, tw);
        if (pl.nsplit > 1) n_part = std::max(n_part, (size_t)pl.tiles * pl.nsplit * 64 * 256);
    }
    MfWs s;
    const int npair = (K + 1) / 2;
    size_t o = 0;
    s.f = o;    o += mf_align((size_t)mid * K * K * sizeof(float));
    s.w1f = o;  o += mf_align((size_t)mid * 64 * 2);
    s.b1 = o;   o += mf_align((size_t)mid * sizeof(float));
    s.taps = o; o += mf_align((size_t)mid * ((K * npair + 1 + 3) / 4 * 4) * sizeof(uint32_t));
    s.b2 = o;   o += mf_align((size_t)mid * sizeof(float));
    s.w2f = o;  o += mf_align((size_t)64 * mid * 2);
    s.b3 = o;   o += mf_align(64 * sizeof(float));
    s.part = o;     o += mf_align(n_part * sizeof(float));
    s.total = o;
    return s;
}

static bool mf_supported(const ofasr_mbconv_desc* d) {
    return d && (d->dtype == OFASR_BF16 || d->dtype == OFASR_F16) && d->Cin == 64 && d->Cout == 64 && d->mid > 0 &&
           d->mid % MF_MC == 0 && (d->K == 3 || d->K == 5 || d->K == 7) && !d->bn_training[0] && !d->bn_training[1] &&
           !d->bn_training[2] && d->N > 0 && d->H > 0 && d->W > 0;
}

static std::atomic<int> g_mf_tile{[] { const char* e = getenv("OFASR_MBFUSED_TILE"); return e ? atoi(e) : 0; }()};

// Tile shape.  Wide tiles move whole 128-byte lines (the 16x16 tile touches 32-byte row pieces, four workgroups per
// line) and measured 7 % (8x32) / 12 % (4x64) faster than 16x16 on images they cover without waste (N=16, mid 384:
// 64x64 43.7 / 55.3 / 70.6 us at k = 3 / 5 / 7 with 4x64 against 49.8 / 64.2 / 76.2 with 16x16); on other sizes the pixels a
// tile hangs over the border are computed for nothing.  Pick the best (covered fraction x speed).
static int mf_pick_tile(int64_t H, int64_t W) {
    static const struct { int tw; double speed; } shapes[3] = {{16, 1.0}, {32, 1.07}, {64, 1.12}};
    int best = 16;
    double best_score = 0.0;
    for (const auto& sh : shapes) {
        const int th = 256 / sh.tw;
        const double covered = (double)(H * W) / ((double)(cdiv(W, sh.tw) * sh.tw) * (double)(cdiv(H, th) * th));
        if (covered * sh.speed > best_score) { best = sh.tw; best_score = covered * sh.speed; }
    }
    return best;
}

template <typename T>
static int mf_fold(const ofasr_mbconv_desc* d, char* ws, const MfWs& s, hipStream_t st) {
    MfFold p;
    p.w1 = d->w1; p.ldw1 = d->ldw1; p.w2 = d->w2; p.ldw2 = d->ldw2;
    p.f = reinterpret_cast<const float*>(ws + s.f);
    for (int i = 0; i < 3; ++i) {
        p.gamma[i] = d->gamma[i]; p.beta[i] = d->beta[i]; p.mean[i] = d->running_mean[i]; p.var[i] = d->running_var[i];
        p.eps[i] = (float)d->bn_eps[i];
    }
    p.mid = (int)d->mid; p.K = d->K;
    T* w1f = reinterpret_cast<T*>(ws + s.w1f);
    T* w2f = reinterpret_cast<T*>(ws + s.w2f);
    float* b1 = reinterpret_cast<float*>(ws + s.b1);
    float* b2 = reinterpret_cast<float*>(ws + s.b2);
    float* b3 = reinterpret_cast<float*>(ws + s.b3);
    uint32_t* taps = reinterpret_cast<uint32_t*>(ws + s.taps);
    OFASR_LAUNCH((mb_fold_kernel<T>), dim3(96), dim3(256), 0, st, p, w1f, b1, taps, b2, w2f, b3);
    return check_launch("ofasr_mbconv_infer_prepare");
}

template <typename T>
static int mf_run(const ofasr_mbconv_desc* d, const void* x, void* out, char* ws, const MfWs& s, float* part,
                  hipStream_t st) {
    T* w1f = reinterpret_cast<T*>(ws + s.w1f);
    T* w2f = reinterpret_cast<T*>(ws + s.w2f);
    float* b1 = reinterpret_cast<float*>(ws + s.b1);
    float* b2 = reinterpret_cast<float*>(ws + s.b2);
    float* b3 = reinterpret_cast<float*>(ws + s.b3);
    uint32_t* taps = reinterpret_cast<uint32_t*>(ws + s.taps);
    int rc = OFASR_OK;
    // OFASR_MBFUSED_TILE=16|32|64 forces the tile width (mf_pick_tile otherwise)
    int tw = g_mf_tile.load(std::memory_order_relaxed);
    if (tw != 16 && tw != 32 && tw != 64) tw = mf_pick_tile(d->H, d->W);
    const MfPlan pl = mf_plan(d, tw);
    const int tiles_x = pl.tiles_x, tiles_y = pl.tiles_y, nsplit = pl.nsplit;
    const long long blocks = pl.tiles * nsplit;
    OFASR_REQUIRE(blocks <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "ofasr_mbconv_infer: too many tiles");
    const double px = (double)d->N * (double)d->H * (double)d->W;
    prof_note(2.0 * px * 64 * (d->residual ? 3.0 : 2.0), 2.0 * px * (2.0 * 64 * d->mid + (double)d->K * d->K * d->mid));
#define OFASR_MF(KK, TH, TW)                                                                                          \
    OFASR_LAUNCH((mb_fused_kernel<T, KK, TH, TW>), dim3((unsigned)blocks), dim3(MF_THREADS), 0, st, (const T*)x,       \
                 (T*)out, (const T*)w1f, (const float*)b1, (const uint32_t*)taps, (const float*)b2, (const T*)w2f,     \
                 (const float*)b3, (int)d->mid, (int)d->H, (int)d->W, tiles_x, tiles_y, d->residual, nsplit, part)
#define OFASR_MFK(TH, TW)                  \
    do {                                   \
        if (d->K == 7) OFASR_MF(7, TH, TW); \
        else if (d->K == 5) OFASR_MF(5, TH, TW); \
        else OFASR_MF(3, TH, TW);          \
    } while (0)
    if (tw == 64) OFASR_MFK(4, 64);
    else if (tw == 32) OFASR_MFK(8, 32);
    else OFASR_MFK(16, 16);
#undef OFASR_MFK
#undef OFASR_MF
    rc = check_launch("ofasr_mbconv_infer");
    if (rc || nsplit == 1) return rc;
#define OFASR_MFS(TH, TW)                                                                                              \
    OFASR_LAUNCH((mb_fused_sum_kernel<T, TH, TW>), dim3((unsigned)(pl.tiles * 8)), dim3(256), 0, st, (const float*)part, \
                 (const T*)x, (T*)out, (const float*)b3, (int)d->H, (int)d->W, tiles_x, tiles_y, d->residual, nsplit)
    if (tw == 64) OFASR_MFS(4, 64);
    else if (tw == 32) OFASR_MFS(8, 32);
    else OFASR_MFS(16, 16);
#undef OFASR_MFS
    return check_launch("ofasr_mbconv_infer");
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT int ofasr_debug_mbfused_tile(int width) { return g_mf_tile.exchange(width, std::memory_order_relaxed); }
OFASR_EXPORT int ofasr_debug_mbfused_split(int enable) { return g_mf_split.exchange(enable ? 1 : 0, std::memory_order_relaxed); }

OFASR_EXPORT int ofasr_mbconv_infer_supported(const ofasr_mbconv_desc* d) { return mf_supported(d) ? 1 : 0; }

OFASR_EXPORT size_t ofasr_mbconv_infer_workspace(const ofasr_mbconv_desc* d) {
    if (!mf_supported(d)) return 0;
    return mf_ws(d).total;
}
OFASR_EXPORT size_t ofasr_mbconv_infer_operand_bytes(const ofasr_mbconv_desc* d) {
    if (!mf_supported(d)) return 0;
    return mf_ws(d).part;
}
OFASR_EXPORT size_t ofasr_mbconv_infer_scratch_bytes(const ofasr_mbconv_desc* d) {
    if (!mf_supported(d)) return 0;
    const MfWs s = mf_ws(d);
    return s.total - s.part;
}

static int mf_check(const char* name, const ofasr_mbconv_desc* d) {
    OFASR_REQUIRE(d != nullptr, OFASR_ERR_INVALID_ARG, "%s: null descriptor", name);
    OFASR_REQUIRE(mf_supported(d), OFASR_ERR_UNSUPPORTED,
                  "%s: needs eval-mode BN, 16-bit activations, 64 -> mid (multiple of 32) -> 64 channels, K in {3,5,7}", name);
    return OFASR_OK;
}

OFASR_EXPORT int ofasr_mbconv_infer_prepare(const ofasr_mbconv_desc* d, void* operands, size_t operand_bytes, void* stream) {
    const char* name = "ofasr_mbconv_infer_prepare";
    int rc = mf_check(name, d);
    if (rc) return rc;
    OFASR_REQUIRE(d->w1 && d->w2 && d->wdw_max, OFASR_ERR_INVALID_ARG, "%s: null weight", name);
    for (int i = 0; i < 3; ++i)
        OFASR_REQUIRE(d->gamma[i] && d->beta[i] && d->running_mean[i] && d->running_var[i], OFASR_ERR_INVALID_ARG,
                      "%s: null BN tensor %d", name, i);
    OFASR_REQUIRE(d->chain_len >= 1 && d->chain_len <= 4 && d->ks[d->chain_len - 1] == d->K, OFASR_ERR_INVALID_ARG,
                  "%s: bad kernel chain", name);
    const MfWs s = mf_ws(d);
    OFASR_REQUIRE(operands && operand_bytes >= s.part, OFASR_ERR_WORKSPACE, "%s: operand buffer %zu B < required %zu B", name,
                  operand_bytes, s.part);
    char* ws = (char*)operands;
    rc = ofasr_ktransform_fwd(d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform,
                              reinterpret_cast<float*>(ws + s.f), d->mid, stream);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (d->dtype == OFASR_F16) return mf_fold<f16_t>(d, ws, s, st);
    return mf_fold<bf16_t>(d, ws, s, st);
}

OFASR_EXPORT int ofasr_mbconv_infer_run(const ofasr_mbconv_desc* d, const void* x, void* out, const void* operands,
                                        size_t operand_bytes, void* scratch, size_t scratch_bytes, void* stream) {
    const char* name = "ofasr_mbconv_infer_run";
    int rc = mf_check(name, d);
    if (rc) return rc;
    OFASR_REQUIRE(x && out, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    const MfWs s = mf_ws(d);
    OFASR_REQUIRE(operands && operand_bytes >= s.part, OFASR_ERR_WORKSPACE, "%s: operand buffer %zu B < required %zu B", name,
                  operand_bytes, s.part);
    OFASR_REQUIRE(s.total == s.part || (scratch && scratch_bytes >= s.total - s.part), OFASR_ERR_WORKSPACE,
                  "%s: scratch %zu B < required %zu B", name, scratch_bytes, s.total - s.part);
    hipStream_t st = as_stream(stream);
    char* ws = (char*)const_cast<void*>(operands);
    if (d->dtype == OFASR_F16) return mf_run<f16_t>(d, x, out, ws, s, (float*)scratch, st);
    return mf_run<bf16_t>(d, x, out, ws, s, (float*)scratch, st);
}

OFASR_EXPORT int ofasr_mbconv_infer(const ofasr_mbconv_desc* d, const void* x, void* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_mbconv_infer";
    int rc = mf_check(name, d);
    if (rc) return rc;
    const MfWs s = mf_ws(d);
    OFASR_REQUIRE(workspace && workspace_bytes >= s.total, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, s.total);
    rc = ofasr_mbconv_infer_prepare(d, workspace, s.part, stream);
    if (rc) return rc;
    return ofasr_mbconv_infer_run(d, x, out, workspace, s.part, (char*)workspace + s.part, s.total - s.part, stream);
}
