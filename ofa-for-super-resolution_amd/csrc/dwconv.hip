// dwconv.hip -- depthwise KxK convolution (stride 1, zero pad K/2) fwd / dgrad / wgrad for gfx950.
//
// Replaces F.conv2d(groups=C) in DynamicSeparableConv2d.forward (reference
// ofa/elastic_nn/modules/dynamic_op.py:73-84) and its autograd (dgrad = same conv with the
// flipped filter, wgrad = per-channel K*K reductions over N*H*W).
//
// Roofline: HBM.  Algorithmic bytes per launch = B * 2 * N*C*H*W (+ 4*C*K*K), B = elem size;
// flops 2*K*K per output, i.e. 2.25 (k3) .. 12.25 (k7) flop/B in fp32 -- left of the fp32 VALU
// ridge (157 TF / 8 TB/s ~ 20 flop/B), but at k=7 the 49 FMAs per output leave < 2x VALU slack, so
// the inner loop is kept to exactly K*K FMAs + (K-1) lane shifts per input row.
//
// Decomposition ("wave strip"): a wave owns 64 adjacent columns of one (n,c) plane -- one NCHW
// image row per lane-row, 256 contiguous bytes per wave load in fp32 -- and walks down the rows.
// Each input row is loaded ONCE; its K column-shifted versions come from lane shuffles (no LDS
// tile), and it is pushed into a ring of K running output rows held in registers (static
// indexing via unroll-by-K).  The K*K filter taps of the plane's channel are wave-uniform and
// live in SGPRs.  For W <= 64 (the MB stack runs at LR resolution: 64x64, 48x48) the zero padding
// falls out of the shuffles (lanes past the row edge hold 0); wider rows are cut into overlapping
// strips of 64 input columns producing 64-2*pad outputs.
#include "ofasr_common.h"

#include <type_traits>

namespace ofasr {

constexpr int DW_WAVES = 4;  // waves per block

// value held by lane (lane + d), 0 when that lane is outside the wave
template <int D>
__device__ __forceinline__ float lane_shift(float v, int lane) {
    if (D == 0) return v;
    const int src = lane + D;
    const float t = __shfl(v, src & 63, 64);
    return (src >= 0 && src < 64) ? t : 0.f;
}

template <int K>
struct Strips {
    static constexpr int PAD = K / 2;
    // W <= 64: one strip, lane == column, every lane produces output.
    // W  > 64: strips of 64 input columns starting at s*(64-2*PAD) - PAD; lanes [PAD, 64-PAD) produce output.
    static __host__ __device__ int count(int W) { return W <= 64 ? 1 : (W + (64 - 2 * PAD) - 1) / (64 - 2 * PAD); }
    static __device__ int col0(int W, int s) { return W <= 64 ? 0 : s * (64 - 2 * PAD) - PAD; }
    static __device__ bool core(int W, int lane) { return W <= 64 ? true : (lane >= PAD && lane < 64 - PAD); }
};

// ------------------------------------------------------------------------------- fwd / dgrad
// unit = (plane, strip, row chunk).  FLIP selects the 180-degree rotated filter (dgrad).
template <typename T, int K, bool FLIP>
__global__ void __launch_bounds__(64 * DW_WAVES) dw_strip_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                                 T* __restrict__ y, int C, int H, int W,
                                                                 int nstrips, int nchunks, int rows_per_chunk,
                                                                 long long units) {
    constexpr int PAD = K / 2;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long unit = (long long)blockIdx.x * DW_WAVES + wave;
    if (unit >= units) return;
    const int chunk = (int)(unit % nchunks);
    const long long u2 = unit / nchunks;
    const int strip = (int)(u2 % nstrips);
    const long long plane = u2 / nstrips;
    const int c = (int)(plane % C);

    float taps[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) taps[e] = f[(long long)c * K * K + (FLIP ? (K * K - 1 - e) : e)];

    const int col = Strips<K>::col0(W, strip) + lane;
    const bool in_col = col >= 0 && col < W;
    const bool out_col = in_col && Strips<K>::core(W, lane);
    const int h0 = chunk * rows_per_chunk;
    const int h1 = min(H, h0 + rows_per_chunk);
    const T* xp = x + plane * (long long)H * W + col;
    T* yp = y + plane * (long long)H * W + col;

    float acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = 0.f;

    const int hstart = h0 - PAD;
    const int niter = (h1 - h0) + 2 * PAD;
    // K input rows are fetched per batch, one batch ahead of the arithmetic (2K independent loads in flight
    // per wave): the kernel is latency-bound otherwise (one 128/256-byte row per load).
    auto load_row = [&](int t) -> float {
        const int hin = hstart + t;
        return (t < niter && hin >= 0 && hin < H && in_col) ? to_float(xp[(long long)hin * W]) : 0.f;
    };
    float xin[K], xnext[K];
#pragma unroll
    for (int u = 0; u < K; ++u) xin[u] = load_row(u);
    for (int base = 0; base < niter; base += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) xnext[u] = load_row(base + K + u);
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = base + u;
            const int hin = hstart + t;
            const float v = xin[u];
            float s[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                switch (j - PAD) {  // compile-time after unrolling
                    case -3: s[j] = lane_shift<-3>(v, lane); break;
                    case -2: s[j] = lane_shift<-2>(v, lane); break;
                    case -1: s[j] = lane_shift<-1>(v, lane); break;
                    case 0: s[j] = v; break;
                    case 1: s[j] = lane_shift<1>(v, lane); break;
                    case 2: s[j] = lane_shift<2>(v, lane); break;
                    default: s[j] = lane_shift<3>(v, lane); break;
                }
            }
            // input row t feeds filter row i of the output row held in slot (u - i) mod K
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float a = acc[(u - i + K) % K];
#pragma unroll
                for (int j = 0; j < K; ++j) a = fmaf(taps[i * K + j], s[j], a);
                acc[(u - i + K) % K] = a;
            }
            // the output row that just received its last (i = K-1) contribution
            const int hout = hin - PAD;
            if (t < niter && hout >= h0 && hout < h1 && out_col)
                yp[(long long)hout * W] = from_float<T>(acc[(u + 1) % K]);
            acc[(u + 1) % K] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < K; ++u) xin[u] = xnext[u];
    }
}

// ------------------------------------------------------------------------------------- wgrad
// unit = (channel, part): the wave walks images n = part, part+nparts, ... of channel c, all strips,
// all rows, keeping K*K lane-partial sums;  df_part[part][c][i][j] = sum dy[h][w-j+pad] * x[h+i-pad][w]
template <typename T, int K>
__global__ void __launch_bounds__(64 * DW_WAVES) dw_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 float* __restrict__ part_out, int N, int C, int H,
                                                                 int W, int nstrips, int nparts, long long units) {
    constexpr int PAD = K / 2;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long unit = (long long)blockIdx.x * DW_WAVES + wave;
    if (unit >= units) return;
    const int part = (int)(unit % nparts);
    const int c = (int)(unit / nparts);

    float acc[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) acc[e] = 0.f;

    for (int n = part; n < N; n += nparts) {
        const long long plane = (long long)n * C + c;
        for (int strip = 0; strip < nstrips; ++strip) {
            const int col = Strips<K>::col0(W, strip) + lane;
            const bool in_col = col >= 0 && col < W;
            const bool own = in_col && Strips<K>::core(W, lane);  // this lane owns x column `col`
            const T* dp = dy + plane * (long long)H * W + col;
            const T* xp = x + plane * (long long)H * W + col;
            float xr[K];  // ring of x rows: slot m mod K holds row (m - PAD)
#pragma unroll
            for (int m = 0; m < K - 1; ++m) {
                const int r = m - PAD;
                xr[m] = (own && r >= 0 && r < H) ? to_float(xp[(long long)r * W]) : 0.f;
            }
            xr[K - 1] = 0.f;
            auto load_x = [&](int h) -> float {   // the new ring row needed at iteration h
                const int rnew = h + PAD;
                return (own && h < H && rnew < H) ? to_float(xp[(long long)rnew * W]) : 0.f;
            };
            auto load_g = [&](int h) -> float { return (h < H && in_col) ? to_float(dp[(long long)h * W]) : 0.f; };
            float xb[K], gb[K], xn_[K], gn_[K];
#pragma unroll
            for (int u = 0; u < K; ++u) {
                xb[u] = load_x(u);
                gb[u] = load_g(u);
            }
            for (int base = 0; base < H; base += K) {
#pragma unroll
                for (int u = 0; u < K; ++u) {   // next batch in flight while this one is consumed
                    xn_[u] = load_x(base + K + u);
                    gn_[u] = load_g(base + K + u);
                }
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    xr[(u + K - 1) % K] = xb[u];
                    const float g = gb[u];
                    float gs[K];  // gs[j](lane) = dy(lane - j + PAD)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        switch (PAD - j) {
                            case -3: gs[j] = lane_shift<-3>(g, lane); break;
                            case -2: gs[j] = lane_shift<-2>(g, lane); break;
                            case -1: gs[j] = lane_shift<-1>(g, lane); break;
                            case 0: gs[j] = g; break;
                            case 1: gs[j] = lane_shift<1>(g, lane); break;
                            case 2: gs[j] = lane_shift<2>(g, lane); break;
                            default: gs[j] = lane_shift<3>(g, lane); break;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < K; ++i) {
                        const float xv = xr[(u + i) % K];  // x row h + i - PAD
#pragma unroll
                        for (int j = 0; j < K; ++j) acc[i * K + j] = fmaf(gs[j], xv, acc[i * K + j]);
                    }
                }
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    xb[u] = xn_[u];
                    gb[u] = gn_[u];
                }
            }
        }
    }
    float* out = part_out + ((long long)part * C + c) * (K * K);
#pragma unroll
    for (int e = 0; e < K * K; ++e) {
        const float s = wave_sum(acc[e]);
        if (lane == 0) out[e] = s;
    }
}

// =================================================================================================
// Vector ("row-group") kernels: 8-16 bytes per lane, one plane slab per wave.
//
// The strip kernels above move one element per lane per row (128-256 B per wave instruction); at the
// MB stack's plane sizes that is latency-bound long before HBM (Little's law: too few bytes in flight).
// Here a lane owns PXL adjacent pixels of a row, G = W/PXL lanes cover one image row, and the gpw = 64/G
// lane groups of a wave take gpw consecutive row chunks (R rows each) of the SAME plane, so
//   * every wave load/store instruction moves gpw full rows (512 B - 1 KiB), K of them in flight per lane;
//   * the K*K taps of the plane's channel are wave-uniform and sit in SGPRs;
//   * column neighbours come from the adjacent lane with 2*PAD shuffles per PXL outputs.
// PXL is 16 bytes' worth of pixels for K <= 3 (bandwidth-bound) and 4 pixels for K >= 5, where the 25-49
// FMAs per output make the kernel VALU-bound and the K*PXL running sums must stay in registers at high
// occupancy.  Requires W % PXL == 0, W/PXL <= 64 and suitably aligned tensors; else the strip kernels run.
// =================================================================================================
template <typename T, int PXL> struct PxIO;   // PXL pixels of type T <-> floats
template <> struct PxIO<float, 4> {
    typedef uint4 raw_t;
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void unpack(const raw_t& v, float* o) {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ raw_t pack(const float* i) {
        return make_uint4(__float_as_uint(i[0]), __float_as_uint(i[1]), __float_as_uint(i[2]), __float_as_uint(i[3]));
    }
};
template <typename T> struct PxIO<T, 8> {   // 16-bit, 16 bytes
    typedef uint4 raw_t;
    static __device__ __forceinline__ raw_t zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void unpack(const raw_t& v, float* o) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T lo, hi;
            lo.v = (uint16_t)(w[i] & 0xffffu);
            hi.v = (uint16_t)(w[i] >> 16);
            o[2 * i] = to_float(lo);
            o[2 * i + 1] = to_float(hi);
        }
    }
    static __device__ __forceinline__ raw_t pack(const float* i) {
        return make_uint4(pack2<T>(i[0], i[1]), pack2<T>(i[2], i[3]), pack2<T>(i[4], i[5]), pack2<T>(i[6], i[7]));
    }
};
template <typename T> struct PxIO<T, 4> {   // 16-bit, 8 bytes
    typedef uint2 raw_t;
    static __device__ __forceinline__ raw_t zero() { return make_uint2(0, 0); }
    static __device__ __forceinline__ void unpack(const raw_t& v, float* o) {
        const uint32_t w[2] = {v.x, v.y};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            T lo, hi;
            lo.v = (uint16_t)(w[i] & 0xffffu);
            hi.v = (uint16_t)(w[i] >> 16);
            o[2 * i] = to_float(lo);
            o[2 * i + 1] = to_float(hi);
        }
    }
    static __device__ __forceinline__ raw_t pack(const float* i) { return make_uint2(pack2<T>(i[0], i[1]), pack2<T>(i[2], i[3])); }
};

template <typename T, int K> struct VecPx { static constexpr int N = (K <= 3 && sizeof(T) == 2) ? 8 : 4; };

// window = [PAD px of the left lane | own PXL px | PAD px of the right lane]; zero past the row ends
template <int PXL, int PAD>
__device__ __forceinline__ void build_window(const float* v, float* win, int lane, bool has_left, bool has_right) {
#pragma unroll
    for (int i = 0; i < PAD; ++i) {
        const float l = __shfl(v[PXL - PAD + i], (lane + 63) & 63, 64);
        const float r = __shfl(v[i], (lane + 1) & 63, 64);
        win[i] = has_left ? l : 0.f;
        win[PAD + PXL + i] = has_right ? r : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PXL; ++i) win[PAD + i] = v[i];
}

// =================================================================================================
// MFMA path for K >= 5 on 16-bit planes of width 32 / 64 and height <= 64 (the MB stack's shapes).
// The vector kernel above is VALU-issue-bound there (4513 VALU instructions per wave against 1568 ideal packed FMAs:
// every SGPR tap costs a v_mov per two packed FMAs).  Here one kernel row is a GEMM on the matrix cores:
//     Out[h][w] = sum_ky  In_ky[h][j] * T_ky[j][w],   In_ky[h][j] = in[h + ky - P][j],  T_ky[j][w] = f[ky][j - w + P]
// A (32 rows h x 16 columns j) is one ds_read_b128 of the plane image in LDS (rows shifted by ky, 8-aligned column
// chunks, zero halo rows); B is the Toeplitz band of the taps, which only depends on (ky, j-chunk): 3*K fragments per
// channel, built once per wave from a zero-padded tap table.  11x the flops of the direct form, on units 16x faster:
// 12*K MFMAs per 64x64 plane.  The taps are rounded to the activation type (what autocast does to a conv weight).
// wave = (channel, group of NPW images); no block-level barrier (each wave owns its LDS plane).
// =================================================================================================
typedef __attribute__((ext_vector_type(16))) float dwm_f32x16;
typedef __attribute__((ext_vector_type(8))) short dwm_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 dwm_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 dwm_f16x8;
template <typename T> struct DwMma;
template <> struct DwMma<bf16_t> {
    static __device__ __forceinline__ dwm_f32x16 run(dwm_s16x8 a, dwm_s16x8 b, dwm_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dwm_bf16x8, a), __builtin_bit_cast(dwm_bf16x8, b),
                                                       c, 0, 0, 0);
    }
};
template <> struct DwMma<f16_t> {
    static __device__ __forceinline__ dwm_f32x16 run(dwm_s16x8 a, dwm_s16x8 b, dwm_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dwm_f16x8, a), __builtin_bit_cast(dwm_f16x8, b), c,
                                                      0, 0, 0);
    }
};
constexpr int DWM_NPW = 4;   // images per wave (the Toeplitz fragments are built once per wave)
constexpr int DWM_PITCH = 144;
// LDS accesses of one wave execute in issue order; this only keeps the COMPILER from moving differently-typed LDS
// accesses across the point (and drains the LDS queue), which is all a wave-private buffer needs
__device__ __forceinline__ void dwm_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <typename T, int K, bool FLIP, bool XF, bool STAT, bool BX = false>
__global__ void __launch_bounds__(64 * DW_WAVES) dw_mfma_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                                T* __restrict__ y, int N, int C, int H, int W,
                                                                int nslabs, InputXf xf, StatOut so, BnFold fold,
                                                                BwdXf bx = BwdXf{}) {
    constexpr int PAD = K / 2;
    constexpr int LROWS = 64 + 2 * PAD;
    // plane image rows of DWM_PITCH bytes: 8 data chunks of 16 B + ONE zero chunk (index 8) that serves as the left pad
    // (chunk -1) and the right pad (chunk 8) of every row; 144 = 16 * 9 (odd) makes the b128 reads of 16 consecutive rows
    // conflict-free without a swizzle, so every fragment address is a lane constant + an immediate
    __shared__ __attribute__((aligned(16))) char planes[DW_WAVES][LROWS * DWM_PITCH];
    __shared__ __attribute__((aligned(16))) char outs[DW_WAVES][64 * 128];   // output image, written back in 16-byte rows
    __shared__ uint16_t tapt[DW_WAVES][K][96];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long unit = (long long)blockIdx.x * DW_WAVES + wave;
    const int c = (int)(unit % C);
    const int n0 = (int)(unit / C) * DWM_NPW;
    if (n0 >= N) return;   // wave-uniform; the kernel has no block barrier
    char* L = planes[wave];
    const int r = lane & 31, hh = lane >> 5;

    // zero the plane image once (the halo rows stay zero) and build the zero-padded tap rows
    for (int i = lane; i < LROWS * (DWM_PITCH / 16); i += 64) *reinterpret_cast<uint4*>(L + i * 16) = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < K * 96; i += 64) tapt[wave][i / 96][i % 96] = 0;
    dwm_lds_fence();
    for (int e = lane; e < K * K; e += 64)
        tapt[wave][e / K][e % K + 40] = from_float<T>(f[(long long)c * K * K + (FLIP ? (K * K - 1 - e) : e)]).v;
    dwm_lds_fence();
    // Toeplitz fragments: lane (column n = r, half hh) holds T[j = d + 8hh + i][w = n] = tap[ky][d + 8hh + i - n + PAD]
    dwm_s16x8 Bf[K][3];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int base = (-8 + 16 * ch) + 8 * hh - r + PAD + 40;
            dwm_s16x8 b;
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = (short)tapt[wave][ky][base + i];
            Bf[ky][ch] = b;
        }

    float xsc = 1.f, xmu = 0.f, xb = 0.f;   // fused BN + ReLU6 of the input plane (wave-uniform)
    if constexpr (XF) {
        if (fold.cp) {
            double s = 0.0, ss = 0.0;
            // 8 partials per lane and request round (clamped index: the loads are in flight together), same order
            for (int q0 = lane; q0 < fold.P; q0 += 8 * 64) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = q0 + 64 * j < fold.P ? q0 + 64 * j : fold.P - 1;
                    v[j] = fold.cp[(long long)c * fold.P + q];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool live = q0 + 64 * j < fold.P;
                    s += live ? (double)v[j].x : 0.0;
                    ss += live ? (double)v[j].y : 0.0;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o, 64);
                ss += __shfl_xor(ss, o, 64);
            }
            const double mean = s / fold.M;
            double var = ss / fold.M - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double invstd = 1.0 / sqrt(var + fold.eps);
            const double gm = fold.gamma ? (double)fold.gamma[c] : 1.0, bt = fold.beta ? (double)fold.beta[c] : 0.0;
            xsc = (float)(gm * invstd);
            xmu = (float)mean;
            xb = (float)bt;
            if (n0 == 0 && lane == 0) {   // image group 0's wave of channel c keeps the statistics
                fold.mean[c] = (float)mean;
                fold.invstd[c] = (float)invstd;
                fold.scale[c] = xsc;
                fold.shift[c] = (float)(bt - mean * gm * invstd);
                if (fold.running_mean) {
                    const double unb = fold.M > 1.0 ? var * fold.M / (fold.M - 1.0) : var;
                    fold.running_mean[c] =
                        (float)((1.0 - fold.momentum) * (double)fold.running_mean[c] + fold.momentum * mean);
                    fold.running_var[c] = (float)((1.0 - fold.momentum) * (double)fold.running_var[c] + fold.momentum * unb);
                }
                if (c == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (fold.counters[i]) *fold.counters[i] += 1;
                }
            }
        } else {
            xsc = xf.scale[c];
            xmu = xf.mean[c];
            xb = fmaf(xmu, xsc, xf.shift[c]);
        }
    }

    float bmu = 0.f, bsc = 1.f, bxb = 0.f, bka = 0.f, bkbi = 0.f;   // BN(+ReLU6) backward of the input plane (BwdXf)
    if constexpr (BX) {
        bmu = bx.mean[c];
        bsc = bx.scale[c];
        bxb = fmaf(bmu, bsc, bx.shift[c]);
        if (bx.fold_partial) bwdxf_fold(bx, c, bsc, n0 == 0 && lane == 0, bka, bkbi);   // wave-uniform; image group 0 publishes
        else {
            bka = bx.ka[c];
            bkbi = bx.kbi[c];
        }
    }

    int abase[2][3];   // byte offset of this lane's fragment chunk for (column block, chunk pair), row r
#pragma unroll
    for (int wb = 0; wb < 2; ++wb)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int cidx = 4 * wb - 1 + 2 * ch + hh;
            const int cz = (cidx < 0 || cidx >= (W >> 3)) ? 8 : cidx;
            abase[wb][ch] = r * DWM_PITCH + cz * 16;
        }
    const int cpr = W >> 3;                 // 16-byte chunks per image row (4 or 8)
    const int nchunks = H * cpr;            // <= 512: at most 8 chunks per lane
    char* O = outs[wave];
    uint4 nxt[8];
    uint4 nxt2[BX ? 8 : 1];   // BwdXf: the pre-BN plane y beside the gradient plane
    auto load_plane = [&](int pl) {
        const long long plane = (long long)(n0 + pl) * C + c;
        const uint4* src = reinterpret_cast<const uint4*>(x + plane * (long long)H * W);
#pragma unroll
        for (int it = 0; it < 8; ++it) {   // branch-free (a chunk past the plane re-reads chunk 0 and is never used): a load
            const int idx = lane + 64 * it;   // under its bounds test waits for every earlier one (s_waitcnt vmcnt(0))
            nxt[it] = src[idx < nchunks ? idx : 0];
        }
        if constexpr (BX) {
            const uint4* src2 = reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(bx.y) + plane * (long long)H * W);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = lane + 64 * it;
                nxt2[it] = src2[idx < nchunks ? idx : 0];
            }
        }
    };
    load_plane(0);
    for (int pl = 0; pl < DWM_NPW && n0 + pl < N; ++pl) {
        const long long plane = (long long)(n0 + pl) * C + c;
        // ---- the plane image: (fused BN + ReLU6), swizzled LDS rows; the next plane's loads go out right after
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + 64 * it;
            if (idx < nchunks) {
                uint4 v = nxt[it];
                if constexpr (XF) {
                    uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T lo, hi;
                        lo.v = (uint16_t)(wv[i] & 0xffffu);
                        hi.v = (uint16_t)(wv[i] >> 16);
                        const float a = fminf(fmaxf(fmaf(to_float(lo) - xmu, xsc, xb), 0.f), 6.f);
                        const float b = fminf(fmaxf(fmaf(to_float(hi) - xmu, xsc, xb), 0.f), 6.f);
                        wv[i] = pack2<T>(a, b);
                    }
                    v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                }
                if constexpr (BX) {   // dy = scale*dz - ka - (y - mean)*kbi, dz = da inside the ReLU6 window of BN(y)
                    uint32_t wv[4] = {v.x, v.y, v.z, v.w};
                    const uint32_t yv[4] = {nxt2[it].x, nxt2[it].y, nxt2[it].z, nxt2[it].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        T lo, hi, ylo, yhi;
                        lo.v = (uint16_t)(wv[i] & 0xffffu);
                        hi.v = (uint16_t)(wv[i] >> 16);
                        ylo.v = (uint16_t)(yv[i] & 0xffffu);
                        yhi.v = (uint16_t)(yv[i] >> 16);
                        const float t0 = to_float(ylo) - bmu, t1 = to_float(yhi) - bmu;
                        const float p0 = fmaf(t0, bsc, bxb), p1 = fmaf(t1, bsc, bxb);
                        const float z0 = (p0 > 0.f && p0 < 6.f) ? to_float(lo) : 0.f;
                        const float z1 = (p1 > 0.f && p1 < 6.f) ? to_float(hi) : 0.f;
                        wv[i] = pack2<T>(fmaf(-t0, bkbi, fmaf(bsc, z0, -bka)), fmaf(-t1, bkbi, fmaf(bsc, z1, -bka)));
                    }
                    v = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                    if (bx.dy_out)   // wave-uniform: the plane's dy, once, for the weight-gradient kernel
                        reinterpret_cast<uint4*>(reinterpret_cast<T*>(bx.dy_out) + plane * (long long)H * W)[idx] = v;
                }
                const int row = idx / cpr, chunk = idx - row * cpr, lr = row + PAD;
                *reinterpret_cast<uint4*>(L + lr * DWM_PITCH + (chunk << 4)) = v;
            }
        }
        dwm_lds_fence();   // the image is complete before any fragment is read
        if (pl + 1 < DWM_NPW && n0 + pl + 1 < N) load_plane(pl + 1);
        float st_s = 0.f, st_q = 0.f;
        for (int hb = 0; hb < H / 32; ++hb) {
            const char* Lh = L + hb * (32 * DWM_PITCH);
#pragma unroll
            for (int wb = 0; wb < 2; ++wb) {
                if (32 * wb >= W) break;   // wave-uniform; unrolled so that abase[wb][ch] is a register, not an indexed array
                dwm_f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        // input columns 32wb - 8 + 16ch + 8hh: chunk 4wb - 1 + 2ch + hh, -1 and 8 -> the zero chunk;
                        // image row 32hb + r + ky - PAD is LDS row 32hb + r + ky: lane base + immediate
                        const dwm_s16x8 a =
                            *reinterpret_cast<const dwm_s16x8*>(Lh + abase[wb][ch] + ky * DWM_PITCH);
                        // transposed product: rows = output columns w (the Toeplitz operand), columns = image rows h
                        acc = DwMma<T>::run(Bf[ky][ch], a, acc);
                    }
                }
                // lane = image row h = 32hb + r; register group q holds columns w = 32wb + 8q + 4hh + (0..3): 8 bytes
                const int h = 32 * hb + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint2 o = make_uint2(pack2<T>(acc[4 * q], acc[4 * q + 1]), pack2<T>(acc[4 * q + 2], acc[4 * q + 3]));
                    const int chunk = 4 * wb + q;
                    *reinterpret_cast<uint2*>(O + h * 128 + ((chunk ^ ((h >> 1) & 7)) << 4) + 8 * hh) = o;
                    if constexpr (STAT) {
                        const uint32_t wv2[2] = {o.x, o.y};
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            T lo, hi;
                            lo.v = (uint16_t)(wv2[i] & 0xffffu);
                            hi.v = (uint16_t)(wv2[i] >> 16);
                            const float a0 = to_float(lo), a1 = to_float(hi);
                            st_s += a0 + a1;
                            st_q = fmaf(a0, a0, fmaf(a1, a1, st_q));
                        }
                    }
                }
            }
        }
        dwm_lds_fence();   // output image complete; every input fragment of this plane has been read
        uint4* dst = reinterpret_cast<uint4*>(y + plane * (long long)H * W);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + 64 * it;
            if (idx < nchunks) {
                const int row = idx / cpr, chunk = idx - row * cpr;
                dst[idx] = *reinterpret_cast<const uint4*>(O + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
            }
        }
        dwm_lds_fence();   // the output image is read back before the next plane writes it
        if constexpr (STAT) {
            const float s = wave_sum(st_s), q = wave_sum(st_q);
            if (lane == 0) {
                so.partial[(long long)c * so.P + (long long)(n0 + pl) * nslabs] = make_float2(s, q);
                for (int sl = 1; sl < nslabs; ++sl)
                    so.partial[(long long)c * so.P + (long long)(n0 + pl) * nslabs + sl] = make_float2(0.f, 0.f);
            }
        }
    }
}

static bool mfma_geom_ok(int64_t H, int64_t W, int K, const void* a, const void* b) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    return K >= 5 && (W == 32 || W == 64) && H >= 32 && H <= 64 && (H % 32) == 0 && (bits & 15) == 0;
}

struct VecGeom {
    int G, gpw, R, nslabs;   // lanes per row, groups per wave, rows per group, slabs per plane
};

// wave = (plane, slab); group g of the wave owns rows [slab*gpw*R + g*R, +R) of that plane
template <typename T, int K, bool FLIP, bool XF = false, bool STAT = false, bool BX = false>
__global__ void __launch_bounds__(64 * DW_WAVES) dw_vec_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                               T* __restrict__ y, int C, int H, int W, VecGeom vg,
                                                               long long nwaves, InputXf xf = InputXf{},
                                                               StatOut so = StatOut{nullptr, 0}, BnFold fold = BnFold{},
                                                               BwdXf bx = BwdXf{}) {
    constexpr int PAD = K / 2;
    constexpr int PXL = VecPx<T, K>::N;
    typedef PxIO<T, PXL> IO;
    typedef typename IO::raw_t raw_t;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long wunit = (long long)blockIdx.x * DW_WAVES + wave;
    if (wunit >= nwaves) return;
    const int slab = (int)(wunit % vg.nslabs);
    const long long plane = wunit / vg.nslabs;
    const int c = (int)(plane % C);   // wave-uniform => the taps are scalar loads

    float taps[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) taps[e] = f[(long long)c * K * K + (FLIP ? (K * K - 1 - e) : e)];
    float bmu = 0.f, bsc = 1.f, bxb = 0.f, bka = 0.f, bkbi = 0.f;   // BN(+ReLU6) backward of the input plane (BwdXf)
    if constexpr (BX) {
        bmu = bx.mean[c];
        bsc = bx.scale[c];
        bxb = fmaf(bmu, bsc, bx.shift[c]);
        if (bx.fold_partial) bwdxf_fold(bx, c, bsc, plane < C && slab == 0 && lane == 0, bka, bkbi);   // wave-uniform
        else {
            bka = bx.ka[c];
            bkbi = bx.kbi[c];
        }
    }
    float xsc = 1.f, xmu = 0.f, xb = 0.f;            // fused BN + ReLU6 of the input plane (wave-uniform)
    if constexpr (XF) {
        if (fold.cp) {
            // the input BN's finalize, per wave: channel c from its P partials (lanes stride, fp64 butterfly: fixed order)
            double s = 0.0, ss = 0.0;
            // 8 partials per lane and request round (clamped index: the loads are in flight together), same order
            for (int q0 = lane; q0 < fold.P; q0 += 8 * 64) {
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = q0 + 64 * j < fold.P ? q0 + 64 * j : fold.P - 1;
                    v[j] = fold.cp[(long long)c * fold.P + q];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool live = q0 + 64 * j < fold.P;
                    s += live ? (double)v[j].x : 0.0;
                    ss += live ? (double)v[j].y : 0.0;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o, 64);
                ss += __shfl_xor(ss, o, 64);
            }
            const double mean = s / fold.M;
            double var = ss / fold.M - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            const double invstd = 1.0 / sqrt(var + fold.eps);
            const double gm = fold.gamma ? (double)fold.gamma[c] : 1.0, bt = fold.beta ? (double)fold.beta[c] : 0.0;
            xsc = (float)(gm * invstd);
            xmu = (float)mean;
            xb = (float)bt;
            if (plane < C && slab == 0 && lane == 0) {   // image 0's wave of channel c keeps the statistics
                fold.mean[c] = (float)mean;
                fold.invstd[c] = (float)invstd;
                fold.scale[c] = xsc;
                fold.shift[c] = (float)(bt - mean * gm * invstd);
                if (fold.running_mean) {
                    const double unb = fold.M > 1.0 ? var * fold.M / (fold.M - 1.0) : var;
                    fold.running_mean[c] =
                        (float)((1.0 - fold.momentum) * (double)fold.running_mean[c] + fold.momentum * mean);
                    fold.running_var[c] = (float)((1.0 - fold.momentum) * (double)fold.running_var[c] + fold.momentum * unb);
                }
                if (c == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (fold.counters[i]) *fold.counters[i] += 1;
                }
            }
        } else {
            xsc = xf.scale[c];
            xmu = xf.mean[c];
            xb = fmaf(xmu, xsc, xf.shift[c]);
        }
    }

    const int g = lane / vg.G, gl = lane - g * vg.G;
    const bool live = g < vg.gpw;                    // lanes past the last full group idle (W does not divide 64*PXL)
    const int h0 = min(H, (slab * vg.gpw + g) * vg.R);
    const int h1 = live ? min(H, h0 + vg.R) : h0;
    const int Wq = W / PXL;
    const raw_t* xp = reinterpret_cast<const raw_t*>(x + plane * (long long)H * W) + gl;
    raw_t* yp = reinterpret_cast<raw_t*>(y + plane * (long long)H * W) + gl;
    const bool has_left = gl > 0, has_right = gl < vg.G - 1;

    float acc[K][PXL];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int p = 0; p < PXL; ++p) acc[i][p] = 0.f;

    float st_s = 0.f, st_q = 0.f;
    const int hstart = h0 - PAD;
    const int niter_max = vg.R + 2 * PAD;            // wave-uniform trip count (shuffles need every lane)
    const int niter = (h1 - h0) + 2 * PAD;
    auto load_row = [&](int t) -> raw_t {
        const int hin = hstart + t;
        return (live && t < niter && hin >= 0 && hin < H) ? xp[(long long)hin * Wq] : IO::zero();
    };
    const raw_t* yp2 = BX ? reinterpret_cast<const raw_t*>(reinterpret_cast<const T*>(bx.y) + plane * (long long)H * W) + gl : nullptr;
    auto load_row2 = [&](int t) -> raw_t {   // BwdXf: the pre-BN plane beside the gradient plane
        const int hin = hstart + t;
        return (BX && live && t < niter && hin >= 0 && hin < H) ? yp2[(long long)hin * Wq] : IO::zero();
    };
    raw_t raw[K];   // software ring: K row loads in flight per lane
    raw_t raw2[BX ? K : 1];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        raw[u] = load_row(u);
        if constexpr (BX) raw2[u] = load_row2(u);
    }
    for (int base = 0; base < niter_max; base += K) {
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int t = base + u;
            const int hin = hstart + t;
            float v[PXL], win[PXL + 2 * PAD + 1];
            IO::unpack(raw[u], v);
            if constexpr (XF) {   // rows outside the image are the convolution's zero padding, not relu6(beta)
                const bool rv = live && t < niter && hin >= 0 && hin < H;
#pragma unroll
                for (int p = 0; p < PXL; ++p) v[p] = rv ? fminf(fmaxf(fmaf(v[p] - xmu, xsc, xb), 0.f), 6.f) : 0.f;
            }
            if constexpr (BX) {   // dy = scale*dz - ka - (y - mean)*kbi; rows outside the image are zero padding
                float yv[PXL];
                IO::unpack(raw2[u], yv);
                const bool rv = live && t < niter && hin >= 0 && hin < H;
#pragma unroll
                for (int p = 0; p < PXL; ++p) {
                    const float tt = yv[p] - bmu, pre = fmaf(tt, bsc, bxb);
                    const float dz = (pre > 0.f && pre < 6.f) ? v[p] : 0.f;
                    v[p] = rv ? fmaf(-tt, bkbi, fmaf(bsc, dz, -bka)) : 0.f;
                }
                raw2[u] = load_row2(t + K);
                // rows [h0, h1) belong to this lane group (halo rows are some other group's): each dy row is stored once
                if (bx.dy_out && rv && hin >= h0 && hin < h1)
                    (reinterpret_cast<raw_t*>(reinterpret_cast<T*>(bx.dy_out) + plane * (long long)H * W) + gl)[(long long)hin * Wq] = IO::pack(v);
            }
            raw[u] = load_row(t + K);
            build_window<PXL, PAD>(v, win, lane, has_left, has_right);
#pragma unroll
            for (int i = 0; i < K; ++i) {
                // input row t feeds output row (t - i) of the group's chunk; rows outside [0, R) are another
                // group's (or nobody's) -- a wave-uniform test, so halo rows cost loads but no FMAs
                if (t - i >= 0 && t - i < vg.R) {
#pragma unroll
                    for (int p = 0; p < PXL; ++p) {
                        float a = acc[(u - i + K) % K][p];
#pragma unroll
                        for (int j = 0; j < K; ++j) a = fmaf(taps[i * K + j], win[p + j], a);
                        acc[(u - i + K) % K][p] = a;
                    }
                }
            }
            const int hout = hin - PAD;
            if (t < niter && hout >= h0 && hout < h1) {
                const raw_t o = IO::pack(acc[(u + 1) % K]);
                yp[(long long)hout * Wq] = o;
                if constexpr (STAT) {   // statistics of the values as stored (rounded to T)
                    float r[PXL];
                    IO::unpack(o, r);
#pragma unroll
                    for (int p = 0; p < PXL; ++p) {
                        st_s += r[p];
                        st_q = fmaf(r[p], r[p], st_q);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < PXL; ++p) acc[(u + 1) % K][p] = 0.f;
        }
    }
    if constexpr (STAT) {   // one (sum, sum of squares) per wave = per (image, slab) of channel c
        const float s = wave_sum(st_s), q = wave_sum(st_q);
        if (lane == 0) so.partial[(long long)c * so.P + (plane / C) * vg.nslabs + slab] = make_float2(s, q);
    }
}

// wgrad: wave = (plane, slab); partial[wave][i][j] = sum_{h in slab, w} dy[h][w-j+PAD] * x[h+i-PAD][w]
template <typename T, int K, bool XF = false>
__global__ void __launch_bounds__(64 * DW_WAVES) dw_wgrad_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                     float* __restrict__ part_out, int H, int W,
                                                                     VecGeom vg, long long nwaves, int C = 1,
                                                                     InputXf xf = InputXf{}) {
    constexpr int PAD = K / 2;
    constexpr int PXL = 4;
    typedef PxIO<T, PXL> IO;
    typedef typename IO::raw_t raw_t;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long wunit = (long long)blockIdx.x * DW_WAVES + wave;
    if (wunit >= nwaves) return;
    const int slab = (int)(wunit % vg.nslabs);
    const long long plane = wunit / vg.nslabs;
    const int g = lane / vg.G, gl = lane - g * vg.G;
    const bool live = g < vg.gpw;
    const int h0 = min(H, (slab * vg.gpw + g) * vg.R);
    const int h1 = live ? min(H, h0 + vg.R) : h0;
    const int Wq = W / PXL;
    const raw_t* dp = reinterpret_cast<const raw_t*>(dy + plane * (long long)H * W) + gl;
    const raw_t* xp = reinterpret_cast<const raw_t*>(x + plane * (long long)H * W) + gl;
    const bool has_left = gl > 0, has_right = gl < vg.G - 1;

    float acc[K * K];
#pragma unroll
    for (int e = 0; e < K * K; ++e) acc[e] = 0.f;
    float xr[K][PXL];   // ring of x rows: slot m mod K holds row h0 - PAD + m
    // ring loads, branch-free: a row outside the image / the group's chunk reads a clamped row and is zeroed when it is
    // unpacked (a load under its bounds test waits for every earlier request -- s_waitcnt vmcnt(0) -- which serialized
    // the ring: one round trip per row instead of K rows in flight)
    auto load_x = [&](int r) -> raw_t { return xp[(long long)(r < 0 ? 0 : (r < H ? r : H - 1)) * Wq]; };
    auto load_g = [&](int h) -> raw_t { return dp[(long long)(h < H ? h : H - 1) * Wq]; };
    auto x_ok = [&](int r) { return live && r >= 0 && r < H; };
    float xsc = 1.f, xmu = 0.f, xb = 0.f;   // fused BN + ReLU6 of the x plane (wave-uniform channel)
    if constexpr (XF) {
        const int c = (int)(plane % C);
        xsc = xf.scale[c];
        xmu = xf.mean[c];
        xb = fmaf(xmu, xsc, xf.shift[c]);
    }
    auto xform = [&](float* v, int r) {   // r = image row of the values; rows outside the image are zero
        const bool rv = x_ok(r);
        if constexpr (XF) {
#pragma unroll
            for (int p = 0; p < PXL; ++p) v[p] = rv ? fminf(fmaxf(fmaf(v[p] - xmu, xsc, xb), 0.f), 6.f) : 0.f;
        } else {
#pragma unroll
            for (int p = 0; p < PXL; ++p) v[p] = rv ? v[p] : 0.f;
        }
    };
#pragma unroll
    for (int m = 0; m < K - 1; ++m) {
        IO::unpack(load_x(h0 - PAD + m), xr[m]);
        xform(xr[m], h0 - PAD + m);
    }
#pragma unroll
    for (int p = 0; p < PXL; ++p) xr[K - 1][p] = 0.f;
    raw_t rawx[K], rawg[K];
#pragma unroll
    for (int u = 0; u < K; ++u) {
        rawx[u] = load_x(h0 + u + PAD);
        rawg[u] = load_g(h0 + u);
    }
    for (int base = 0; base < vg.R; base += K) {     // wave-uniform trip count
#pragma unroll
        for (int u = 0; u < K; ++u) {
            const int h = h0 + base + u;
            float gv[PXL], gwin[PXL + 2 * PAD + 1];
            IO::unpack(rawx[u], xr[(u + K - 1) % K]);   // row h + PAD enters the ring
            xform(xr[(u + K - 1) % K], h + PAD);
            IO::unpack(rawg[u], gv);
            if (h >= h1) {
#pragma unroll
                for (int p = 0; p < PXL; ++p) gv[p] = 0.f;
            }
            rawx[u] = load_x(h + K + PAD);
            rawg[u] = load_g(h + K);
            build_window<PXL, PAD>(gv, gwin, lane, has_left, has_right);
            if (base + u < vg.R) {   // wave-uniform: iterations past the chunk only drain the unrolled ring
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        float a = acc[i * K + j];
#pragma unroll
                        for (int p = 0; p < PXL; ++p) a = fmaf(xr[(u + i) % K][p], gwin[p + K - 1 - j], a);
                        acc[i * K + j] = a;
                    }
            }
        }
    }
    float* out = part_out + wunit * (K * K);
#pragma unroll
    for (int e = 0; e < K * K; ++e) {
        const float s = wave_sum(acc[e]);
        if (lane == 0) out[e] = s;
    }
}

// df[c][e] = sum_{n, slab} partial[(n*C + c)*nslabs + slab][e]   (fixed order => deterministic)
__global__ void __launch_bounds__(256) dw_wgrad_vec_reduce_kernel(const float* __restrict__ part, float* __restrict__ df,
                                                                  int N, int C, int nslabs, int KK) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)C * KK) return;
    const int c = (int)(idx / KK), e = (int)(idx - (long long)c * KK);
    float s = 0.f;
    for (int n = 0; n < N; ++n)
        for (int q = 0; q < nslabs; ++q) s += part[(((long long)n * C + c) * nslabs + q) * KK + e];
    df[idx] = s;
}

// PXL = pixels per lane the kernel for (dtype, K) uses; ok = the vector path applies
static bool vec_geom(int64_t H, int64_t W, int esize, int PXL, const void* a, const void* b, VecGeom& vg) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    if (H <= 0 || W <= 0 || W % PXL != 0 || W / PXL > 64 || (bits & 15) != 0) return false;
    (void)esize;
    vg.G = (int)(W / PXL);
    vg.gpw = 64 / vg.G;
    int64_t R = cdiv(H, vg.gpw);
    if (R < 4) R = 4;
    if (R > 32) R = 32;
    vg.R = (int)R;
    vg.nslabs = (int)cdiv(H, (int64_t)vg.gpw * vg.R);
    return true;
}

// df[c][e] = sum_part part_out[part][c][e]   (fixed order => deterministic)
__global__ void __launch_bounds__(256) dw_wgrad_reduce_kernel(const float* __restrict__ part_out,
                                                              float* __restrict__ df, int nparts, long long CKK) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CKK) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part_out[(long long)p * CKK + idx];
    df[idx] = s;
}

static int wgrad_parts(int64_t N, int64_t C) {
    // enough waves to fill 256 CUs x ~8 waves, at most one part per image
    int64_t want = cdiv(6144, C > 0 ? C : 1);
    if (want < 1) want = 1;
    if (want > N) want = N;
    return (int)want;
}

template <typename T, bool FLIP, bool XF = false, bool STAT = false, bool BX = false>
static int launch_conv(const char* name, const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H,
                       int64_t W, int K, hipStream_t st, InputXf xf = InputXf{}, StatOut so = StatOut{nullptr, 0},
                       BnFold fold = BnFold{}, BwdXf bx = BwdXf{}) {
    prof_note(2.0 * sizeof(T) * (double)N * (double)C * (double)H * (double)W + 4.0 * (double)C * K * K,
              2.0 * K * K * (double)N * (double)C * (double)H * (double)W);
    if constexpr (sizeof(T) == 2) {
        if (mfma_geom_ok(H, W, K, x, y)) {
            VecGeom vg0;
            const int nslabs = vec_geom(H, W, 2, 4, x, y, vg0) ? vg0.nslabs : 1;   // statistics slot layout of the vector path
            if (STAT && so.P != (int)(N * nslabs)) {
                set_error("%s: statistics slab count %d != %lld", name, so.P, (long long)(N * nslabs));
                return OFASR_ERR_INVALID_ARG;
            }
            const long long units = (long long)C * cdiv(N, DWM_NPW);
            if (K == 5)
                OFASR_LAUNCH((dw_mfma_kernel<T, 5, FLIP, XF, STAT, BX>), dim3((unsigned)cdiv(units, DW_WAVES)),
                                   dim3(64 * DW_WAVES), 0, st, (const T*)x, f, (T*)y, (int)N, (int)C, (int)H, (int)W, nslabs,
                                   xf, so, fold, bx);
            else
                OFASR_LAUNCH((dw_mfma_kernel<T, 7, FLIP, XF, STAT, BX>), dim3((unsigned)cdiv(units, DW_WAVES)),
                                   dim3(64 * DW_WAVES), 0, st, (const T*)x, f, (T*)y, (int)N, (int)C, (int)H, (int)W, nslabs,
                                   xf, so, fold, bx);
            return check_launch(name);
        }
    }
    {
        VecGeom vg;
#define OFASR_DWV(KK)                                                                                               \
    if (vec_geom(H, W, (int)sizeof(T), VecPx<T, KK>::N, x, y, vg)) {                                               \
        const long long nwaves = (long long)N * C * vg.nslabs;                                                     \
        if (STAT && so.P != (int)(N * vg.nslabs)) {                                                                \
            set_error("%s: statistics slab count %d != %lld", name, so.P, (long long)(N * vg.nslabs));             \
            return OFASR_ERR_INVALID_ARG;                                                                          \
        }                                                                                                          \
        OFASR_LAUNCH((dw_vec_kernel<T, KK, FLIP, XF, STAT, BX>), dim3((unsigned)cdiv(nwaves, DW_WAVES)),     \
                           dim3(64 * DW_WAVES), 0, st, (const T*)x, f, (T*)y, (int)C, (int)H, (int)W, vg, nwaves,  \
                           xf, so, fold, bx);                                                                      \
        return check_launch(name);                                                                                 \
    }
        switch (K) {
            case 1: OFASR_DWV(1) break;
            case 3: OFASR_DWV(3) break;
            case 5: OFASR_DWV(5) break;
            default: OFASR_DWV(7) break;
        }
#undef OFASR_DWV
    }
    if constexpr (XF || BX) {
        set_error("%s: fused input transform needs the vector kernel", name);
        return OFASR_ERR_UNSUPPORTED;
    }
    const int rows_per_chunk = H <= 32 ? (int)H : 32;
    const int nchunks = (int)cdiv(H, rows_per_chunk);
#define OFASR_DW_LAUNCH(KK)                                                                                  \
    {                                                                                                        \
        const int nstrips = Strips<KK>::count((int)W);                                                       \
        const long long units = (long long)N * C * nstrips * nchunks;                                        \
        OFASR_LAUNCH((dw_strip_kernel<T, KK, FLIP>), dim3((unsigned)cdiv(units, DW_WAVES)),            \
                           dim3(64 * DW_WAVES), 0, st, (const T*)x, f, (T*)y, (int)C, (int)H, (int)W, nstrips, \
                           nchunks, rows_per_chunk, units);                                                  \
    }
    switch (K) {
        case 1: OFASR_DW_LAUNCH(1); break;
        case 3: OFASR_DW_LAUNCH(3); break;
        case 5: OFASR_DW_LAUNCH(5); break;
        default: OFASR_DW_LAUNCH(7); break;
    }
#undef OFASR_DW_LAUNCH
    return check_launch(name);
}

static int check_conv_args(const char* name, const void* a, const void* b, const void* c, int64_t N, int64_t C,
                           int64_t H, int64_t W, int K, int dtype) {
    OFASR_REQUIRE(a && b && c, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0, OFASR_ERR_INVALID_ARG, "%s: negative size", name);
    OFASR_REQUIRE(K == 1 || K == 3 || K == 5 || K == 7, OFASR_ERR_UNSUPPORTED, "%s: K=%d not in {1,3,5,7}", name, K);
    OFASR_REQUIRE(dtype == OFASR_F32 || dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_INVALID_ARG,
                  "%s: bad dtype %d", name, dtype);
    OFASR_REQUIRE(C <= INT32_MAX && H <= INT32_MAX && W <= INT32_MAX && N <= INT32_MAX, OFASR_ERR_UNSUPPORTED,
                  "%s: dimension too large", name);
    return OFASR_OK;
}

template <bool FLIP>
static int conv_entry(const char* name, const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H,
                      int64_t W, int K, int dtype, void* stream) {
    int rc = check_conv_args(name, x, f, y, N, C, H, W, K, dtype);
    if (rc) return rc;
    if (N * C * H * W == 0) return OFASR_OK;
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case OFASR_F32: return launch_conv<float, FLIP>(name, x, f, y, N, C, H, W, K, st);
        case OFASR_F16: return launch_conv<f16_t, FLIP>(name, x, f, y, N, C, H, W, K, st);
        default: return launch_conv<bf16_t, FLIP>(name, x, f, y, N, C, H, W, K, st);
    }
}

// =================================================================================================
// Depthwise weight gradient on the matrix cores (16-bit activations, W in {32, 64}, H a multiple of 16 up to 64).
//     dW[ky][kx] = sum_{h,w} dY[h][w] * A[h + ky - P][w + kx - P]
// For a kernel row ky the products of ALL column pairs are one GEMM over the image rows,
//     M_ky[j][w] = sum_h A[h + ky - P][j] * dY[h][w]            (j input column, w output column),
// and dW[ky][kx] is the sum of M_ky along the diagonal j - w = kx - P.  One workgroup owns a channel: wave ky keeps the
// W x W accumulator tiles of its kernel row in registers across ALL images of the batch (the planes stream through a
// double-buffered LDS image, both operands are transposing reads of row-major planes, the kernel-row shift is a row
// offset into the plane image with P zero rows above and below), and the diagonals are summed ONCE at the end -- the
// per-lane select-and-add that made a per-plane GEMM formulation no cheaper than the vector kernel is amortised over
// the batch.  No split-K partials, no reduce launch: the kernel writes dW[c] itself.  16 * W/32 * W/32 MFMAs per
// (image, kernel row) on 64 image rows; the planes are read once (2 tensors), i.e. the kernel is HBM-bound
// (dw_wgrad_vec_kernel<7>: 4513 VALU instructions per wave against 1568 ideal, 113 us at N=16, 384 planes of 64x64).
typedef __attribute__((ext_vector_type(4))) short dwg_s16x4;
typedef __attribute__((address_space(3))) dwg_s16x4 dwg_lds_s16x4;
constexpr int DWG_PITCH = 96;      // pixels per plane row in LDS: 192 bytes = 192 (mod 256), so the four rows of a
                                   // transposing read sit on disjoint banks
constexpr int DWG_THREADS = 512;

// 8 k-values (plane rows kb .. kb+7) of plane column pos0 + (lane & 31): the A (row = column index) and B (column =
// column index) operand layout of the 32x32x16 MFMA, by two transposing reads
__device__ __forceinline__ dwm_s16x8 dwg_frag(const char* img, int pos0, int s, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kb = 16 * s + 8 * (g >> 1);
    const int colb = (pos0 + 16 * (g & 1) + 4 * pp) * 2;
    const dwg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dwg_lds_s16x4*)(img + (kb + q) * (DWG_PITCH * 2) + colb));
    const dwg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((dwg_lds_s16x4*)(img + (kb + 4 + q) * (DWG_PITCH * 2) + colb));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// GBX: the gradient operand is formed from (da, y) through the BN(+ReLU6) backward as it is staged (BwdXf) -- the depthwise
// input gradient on the chain then does not have to store dy for this kernel (one tensor less written on the chain, one
// more read here, on the side stream)
template <typename T, int K, int WT, bool XF, bool GBX = false>
__global__ void __launch_bounds__(DWG_THREADS) dw_wgrad_mfma_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                    float* __restrict__ df, int N, int C, int H,
                                                                    InputXf xf, BwdXf bx = BwdXf{}) {
    constexpr int P = K / 2, W = 32 * WT;
    constexpr int AROWS = 64 + 2 * P;
    constexpr int A_BYTES = AROWS * DWG_PITCH * 2, G_BYTES = 64 * DWG_PITCH * 2;
    __shared__ __attribute__((aligned(16))) char lds[2][A_BYTES + G_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x;
    float mu = 0.f, sc = 1.f, be = 0.f;
    if constexpr (XF) {
        mu = xf.mean[c];
        sc = xf.scale[c];
        be = fmaf(mu, sc, xf.shift[c]);
    }
    float bmu = 0.f, bsc = 1.f, bxb = 0.f, bka = 0.f, bkbi = 0.f;
    if constexpr (GBX) {
        bmu = bx.mean[c];
        bsc = bx.scale[c];
        bxb = fmaf(bmu, bsc, bx.shift[c]);
        bka = bx.ka[c];
        bkbi = bx.kbi[c];
    }
    // zero halo rows of both A images (rows [0, P) and [P + H, P + H + P)); the data rows are rewritten per image
    for (int e = tid; e < 2 * 2 * P * (DWG_PITCH / 8); e += DWG_THREADS) {
        const int b = e / (2 * P * (DWG_PITCH / 8)), r2 = (e / (DWG_PITCH / 8)) % (2 * P), q = e % (DWG_PITCH / 8);
        const int row = r2 < P ? r2 : P + H + (r2 - P);
        *reinterpret_cast<uint4*>(lds[b] + row * (DWG_PITCH * 2) + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    const int quads = H * (W / 8);                 // 16-byte pieces per plane (<= 512)
    const bool mine = tid < quads;
    const int row = tid / (W / 8), col8 = tid - row * (W / 8);
    const long long plane = (long long)H * W;
    uint4 ga = make_uint4(0u, 0u, 0u, 0u), gg = ga, gy = ga;
    auto load = [&](int n) {
        if (mine) {
            const long long off = ((long long)n * C + c) * plane + (long long)row * W + 8 * col8;
            ga = *reinterpret_cast<const uint4*>(x + off);
            gg = *reinterpret_cast<const uint4*>(dy + off);
            if constexpr (GBX) gy = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(bx.y) + off);
        }
    };
    auto store = [&](int b) {
        if (mine) {
            uint4 a = ga;
            if constexpr (XF) {   // the activated operand relu6(BN(y)) as the matrix cores see it (16-bit)
                const uint32_t w[4] = {ga.x, ga.y, ga.z, ga.w};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float lo, hi;
                    unpack2<T>(w[i], lo, hi);
                    o[i] = pack2<T>(fminf(fmaxf(fmaf(lo - mu, sc, be), 0.f), 6.f), fminf(fmaxf(fmaf(hi - mu, sc, be), 0.f), 6.f));
                }
                a = make_uint4(o[0], o[1], o[2], o[3]);
            }
            uint4 g = gg;
            if constexpr (GBX) {   // dy = scale*dz - ka - (y - mean)*kbi, dz = da inside the ReLU6 window of BN(y): the arithmetic
                                   // of dw_mfma_kernel's BX staging, so the weight gradient sees the dy the input gradient saw
                const uint32_t wv[4] = {gg.x, gg.y, gg.z, gg.w}, yv[4] = {gy.x, gy.y, gy.z, gy.w};
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float d0, d1, y0, y1;
                    unpack2<T>(wv[i], d0, d1);
                    unpack2<T>(yv[i], y0, y1);
                    const float t0 = y0 - bmu, t1 = y1 - bmu;
                    const float p0 = fmaf(t0, bsc, bxb), p1 = fmaf(t1, bsc, bxb);
                    const float z0 = (p0 > 0.f && p0 < 6.f) ? d0 : 0.f, z1 = (p1 > 0.f && p1 < 6.f) ? d1 : 0.f;
                    o[i] = pack2<T>(fmaf(-t0, bkbi, fmaf(bsc, z0, -bka)), fmaf(-t1, bkbi, fmaf(bsc, z1, -bka)));
                }
                g = make_uint4(o[0], o[1], o[2], o[3]);
            }
            *reinterpret_cast<uint4*>(lds[b] + (row + P) * (DWG_PITCH * 2) + col8 * 16) = a;
            *reinterpret_cast<uint4*>(lds[b] + A_BYTES + row * (DWG_PITCH * 2) + col8 * 16) = g;
        }
    };
    dwm_f32x16 acc[WT][WT];
#pragma unroll
    for (int jb = 0; jb < WT; ++jb)
#pragma unroll
        for (int wb = 0; wb < WT; ++wb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[jb][wb][i] = 0.f;
    const int ksteps = H / 16;
    load(0);
    for (int n = 0; n < N; ++n) {
        const int b = n & 1;
        store(b);
        __syncthreads();      // image n visible; every wave is done with image n - 1 (the buffer image n + 1 will take)
        if (n + 1 < N) load(n + 1);
        if (wave < K) {
            const char* ab = lds[b] + wave * (DWG_PITCH * 2);      // kernel row ky = wave: plane rows shifted by ky
            const char* gb = lds[b] + A_BYTES;
            for (int s = 0; s < ksteps; ++s) {
                dwm_s16x8 af[WT], bf[WT];
#pragma unroll
                for (int jb = 0; jb < WT; ++jb) af[jb] = dwg_frag(ab, 32 * jb, s, lane);
#pragma unroll
                for (int wb = 0; wb < WT; ++wb) bf[wb] = dwg_frag(gb, 32 * wb, s, lane);
#pragma unroll
                for (int jb = 0; jb < WT; ++jb)
#pragma unroll
                    for (int wb = 0; wb < WT; ++wb) acc[jb][wb] = DwMma<T>::run(af[jb], bf[wb], acc[jb][wb]);
            }
        }
    }
    if (wave >= K) return;
    // diagonals: accumulator (row r, column cc) of tile (jb, wb) is M[j = 32 jb + r][w = 32 wb + cc]; kx = j - w + P
    float sum[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) sum[kx] = 0.f;
    const int cc = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int jb = 0; jb < WT; ++jb)
#pragma unroll
        for (int wb = 0; wb < WT; ++wb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                const int kx = 32 * (jb - wb) + r - cc + P;
                const float v = acc[jb][wb][reg];
#pragma unroll
                for (int q = 0; q < K; ++q) sum[q] += (kx == q) ? v : 0.f;
            }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
        const float t = wave_sum(sum[kx]);
        if (lane == 0) df[((long long)c * K + wave) * K + kx] = t;
    }
}

static bool dw_wgrad_mfma_ok(const void* dy, const void* x, int64_t N, int64_t C, int64_t H, int64_t W, int K, size_t es) {
    static const bool on = [] { const char* e = getenv("OFASR_DW_WGRAD_MFMA"); return !(e && e[0] == '0'); }();
    return on && es == 2 && (K == 3 || K == 5 || K == 7) && (W == 32 || W == 64) && H % 16 == 0 && H >= 16 && H <= 64 &&
           N >= 1 && C >= 1 && C <= INT32_MAX && N <= INT32_MAX &&
           ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
}

template <typename T, bool XF = false>
static int launch_wgrad(const char* name, const void* dy, const void* x, float* df, int64_t N, int64_t C,
                        int64_t H, int64_t W, int K, float* ws, hipStream_t st, InputXf xf = InputXf{}) {
    prof_note(2.0 * sizeof(T) * (double)N * (double)C * (double)H * (double)W + 4.0 * (double)C * K * K,
              2.0 * K * K * (double)N * (double)C * (double)H * (double)W);
    if constexpr (!std::is_same<T, float>::value) {
        if (dw_wgrad_mfma_ok(dy, x, N, C, H, W, K, sizeof(T))) {
#define OFASR_DWGM(KK, WT)                                                                                          \
    OFASR_LAUNCH((dw_wgrad_mfma_kernel<T, KK, WT, XF>), dim3((unsigned)C), dim3(DWG_THREADS), 0, st, (const T*)dy,   \
                 (const T*)x, df, (int)N, (int)C, (int)H, xf)
            if (W == 64) {
                if (K == 7) OFASR_DWGM(7, 2); else if (K == 5) OFASR_DWGM(5, 2); else OFASR_DWGM(3, 2);
            } else {
                if (K == 7) OFASR_DWGM(7, 1); else if (K == 5) OFASR_DWGM(5, 1); else OFASR_DWGM(3, 1);
            }
#undef OFASR_DWGM
            return check_launch(name);
        }
    }
    {
        VecGeom vg;
        if (vec_geom(H, W, (int)sizeof(T), 4, dy, x, vg)) {
            const long long nwaves = (long long)N * C * vg.nslabs;
            const unsigned grid = (unsigned)cdiv(nwaves, DW_WAVES);
#define OFASR_DWWV(KK)                                                                                              \
    OFASR_LAUNCH((dw_wgrad_vec_kernel<T, KK, XF>), dim3(grid), dim3(64 * DW_WAVES), 0, st, (const T*)dy,      \
                       (const T*)x, ws, (int)H, (int)W, vg, nwaves, (int)C, xf)
            switch (K) {
                case 1: OFASR_DWWV(1); break;
                case 3: OFASR_DWWV(3); break;
                case 5: OFASR_DWWV(5); break;
                default: OFASR_DWWV(7); break;
            }
#undef OFASR_DWWV
            int rc0 = check_launch(name);
            if (rc0) return rc0;
            const long long CKK0 = (long long)C * K * K;
            OFASR_LAUNCH(dw_wgrad_vec_reduce_kernel, dim3((unsigned)cdiv(CKK0, 256)), dim3(256), 0, st, ws, df,
                               (int)N, (int)C, vg.nslabs, K * K);
            return check_launch(name);
        }
    }
    if constexpr (XF) {
        set_error("%s: fused input transform needs the vector kernel", name);
        return OFASR_ERR_UNSUPPORTED;
    }
    const int nparts = wgrad_parts(N, C);
    const long long units = (long long)C * nparts;
#define OFASR_DW_WG(KK)                                                                                        \
    OFASR_LAUNCH((dw_wgrad_kernel<T, KK>), dim3((unsigned)cdiv(units, DW_WAVES)), dim3(64 * DW_WAVES), 0, \
                       st, (const T*)dy, (const T*)x, ws, (int)N, (int)C, (int)H, (int)W,                      \
                       Strips<KK>::count((int)W), nparts, units)
    switch (K) {
        case 1: OFASR_DW_WG(1); break;
        case 3: OFASR_DW_WG(3); break;
        case 5: OFASR_DW_WG(5); break;
        default: OFASR_DW_WG(7); break;
    }
#undef OFASR_DW_WG
    int rc = check_launch(name);
    if (rc) return rc;
    const long long CKK = (long long)C * K * K;
    OFASR_LAUNCH(dw_wgrad_reduce_kernel, dim3((unsigned)cdiv(CKK, 256)), dim3(256), 0, st, ws, df, nparts,
                       CKK);
    return check_launch(name);
}

bool dwconv_xf_supported(const void* x, const void* y, int64_t H, int64_t W, int K, int dtype) {
    if (dtype != OFASR_F16 && dtype != OFASR_BF16) return false;
    VecGeom vg;
    const int pxl = K <= 3 ? 8 : 4;
    return vec_geom(H, W, 2, pxl, x, y, vg) && vec_geom(H, W, 2, 4, x, y, vg);
}

int dwconv_stat_units(int64_t N, int64_t H, int64_t W, int K, int dtype) {
    VecGeom vg;
    const int pxl = (K <= 3 && dtype != OFASR_F32) ? 8 : 4;
    static const char dummy[16] __attribute__((aligned(16))) = {0};
    if (!vec_geom(H, W, 2, pxl, dummy, dummy, vg)) return 0;
    return (int)(N * vg.nslabs);
}

// plain depthwise forward whose kernel also leaves the statistics partials of y ([C][N * slabs] (sum, sum of squares),
// dwconv_stat_units) -- any element type, where the vector kernel applies (the un-fused composite path: fp32)
bool dwconv_stat_supported(const void* x, const void* y, int64_t H, int64_t W, int K, int dtype) {
    VecGeom vg;
    const int pxl = (K <= 3 && dtype != OFASR_F32) ? 8 : 4;
    if (dtype != OFASR_F32 && mfma_geom_ok(H, W, K, x, y)) return vec_geom(H, W, 2, 4, x, y, vg);
    return vec_geom(H, W, dtype == OFASR_F32 ? 4 : 2, pxl, x, y, vg);
}
int dwconv_fwd_stat(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H, int64_t W, int K, int dtype,
                    StatOut so, void* stream) {
    const char* name = "dwconv_fwd_stat";
    int rc = check_conv_args(name, x, f, y, N, C, H, W, K, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(so.partial != nullptr && dwconv_stat_supported(x, y, H, W, K, dtype), OFASR_ERR_UNSUPPORTED,
                  "%s: needs the vector kernel and a statistics buffer", name);
    if (N * C * H * W == 0) return OFASR_OK;
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F32) return launch_conv<float, false, false, true>(name, x, f, y, N, C, H, W, K, st, InputXf{}, so);
    if (dtype == OFASR_F16) return launch_conv<f16_t, false, false, true>(name, x, f, y, N, C, H, W, K, st, InputXf{}, so);
    return launch_conv<bf16_t, false, false, true>(name, x, f, y, N, C, H, W, K, st, InputXf{}, so);
}

int dwconv_fwd_xf(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H, int64_t W, int K, int dtype,
                  InputXf xf, void* stream, StatOut so, BnFold fold) {
    const char* name = "dwconv_fwd_xf";
    int rc = check_conv_args(name, x, f, y, N, C, H, W, K, dtype);
    if (rc) return rc;
    OFASR_REQUIRE((xf.scale && xf.shift && xf.mean) ||
                      (fold.cp && fold.P > 0 && fold.mean && fold.invstd && fold.scale && fold.shift),
                  OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    if (N * C * H * W == 0) return OFASR_OK;
    hipStream_t st = as_stream(stream);
    if (so.partial) {
        if (dtype == OFASR_F16)
            return launch_conv<f16_t, false, true, true>(name, x, f, y, N, C, H, W, K, st, xf, so, fold);
        return launch_conv<bf16_t, false, true, true>(name, x, f, y, N, C, H, W, K, st, xf, so, fold);
    }
    if (dtype == OFASR_F16)
        return launch_conv<f16_t, false, true>(name, x, f, y, N, C, H, W, K, st, xf, StatOut{nullptr, 0}, fold);
    return launch_conv<bf16_t, false, true>(name, x, f, y, N, C, H, W, K, st, xf, StatOut{nullptr, 0}, fold);
}

int dwconv_wgrad_xf(const void* dy, const void* x, float* df, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                    int dtype, InputXf xf, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "dwconv_wgrad_xf";
    int rc = check_conv_args(name, dy, x, df, N, C, H, W, K, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(xf.scale && xf.shift && xf.mean, OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    OFASR_REQUIRE(N * C * H * W > 0, OFASR_ERR_UNSUPPORTED, "%s: empty tensor", name);
    const size_t need = ofasr_dwconv_wgrad_workspace(N, C, H, W, K);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16) return launch_wgrad<f16_t, true>(name, dy, x, df, N, C, H, W, K, (float*)workspace, st, xf);
    return launch_wgrad<bf16_t, true>(name, dy, x, df, N, C, H, W, K, (float*)workspace, st, xf);
}

// weight gradient on the matrix cores with BOTH operands formed on read: a = relu6(BN(x)) (InputXf) and the gradient
// dy(da, y) through the BN backward (BwdXf).  Only where dw_wgrad_mfma_kernel applies (dwconv_wgrad_bx_supported).
bool dwconv_wgrad_bx_supported(const void* da, const void* x, const void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                               int K, int dtype) {
    static const bool on = [] { const char* e = getenv("OFASR_DW_WGRAD_BX"); return !(e && e[0] == '0'); }();
    return on && (dtype == OFASR_F16 || dtype == OFASR_BF16) && dw_wgrad_mfma_ok(da, x, N, C, H, W, K, 2) &&
           (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

int dwconv_wgrad_xf_bx(const void* da, const void* x, float* df, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                       int dtype, InputXf xf, BwdXf bx, void* stream) {
    const char* name = "dwconv_wgrad_xf_bx";
    OFASR_REQUIRE(da && x && df, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(xf.scale && xf.shift && xf.mean && bx.y && bx.mean && bx.scale && bx.shift && bx.ka && bx.kbi,
                  OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dwconv_wgrad_bx_supported(da, x, bx.y, N, C, H, W, K, dtype), OFASR_ERR_UNSUPPORTED,
                  "%s: shape outside the matrix-core weight-gradient kernel", name);
    hipStream_t st = as_stream(stream);
    prof_note(2.0 * 3.0 * (double)N * (double)C * (double)H * (double)W + 4.0 * (double)C * K * K,
              2.0 * K * K * (double)N * (double)C * (double)H * (double)W);
#define OFASR_DWGB(TT, KK, WT)                                                                                       \
    OFASR_LAUNCH((dw_wgrad_mfma_kernel<TT, KK, WT, true, true>), dim3((unsigned)C), dim3(DWG_THREADS), 0, st,         \
                 (const TT*)da, (const TT*)x, df, (int)N, (int)C, (int)H, xf, bx)
#define OFASR_DWGB_T(TT)                                                                                             \
    do {                                                                                                            \
        if (W == 64) {                                                                                              \
            if (K == 7) OFASR_DWGB(TT, 7, 2); else if (K == 5) OFASR_DWGB(TT, 5, 2); else OFASR_DWGB(TT, 3, 2);       \
        } else {                                                                                                    \
            if (K == 7) OFASR_DWGB(TT, 7, 1); else if (K == 5) OFASR_DWGB(TT, 5, 1); else OFASR_DWGB(TT, 3, 1);       \
        }                                                                                                           \
    } while (0)
    if (dtype == OFASR_F16) OFASR_DWGB_T(f16_t);
    else OFASR_DWGB_T(bf16_t);
#undef OFASR_DWGB_T
#undef OFASR_DWGB
    return check_launch(name);
}

// input gradient of the depthwise conv with the gradient operand read through the BN(+ReLU6) backward (BwdXf):
// dx = dwconv_dgrad(dy(da, y), f); 16-bit, vector / matrix-core shapes only (dwconv_xf_supported)
int dwconv_dgrad_bx(const void* da, const float* f, void* dx, int64_t N, int64_t C, int64_t H, int64_t W, int K,
                    int dtype, BwdXf bx, void* stream) {
    const char* name = "dwconv_dgrad_bx";
    int rc = check_conv_args(name, da, f, dx, N, C, H, W, K, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(bx.y && bx.mean && bx.scale && bx.shift &&
                      ((bx.ka && bx.kbi) || (bx.fold_partial && bx.fold_invstd && bx.fold_P > 0 && bx.fold_C == C)),
                  OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    OFASR_REQUIRE(((reinterpret_cast<uintptr_t>(bx.y) | reinterpret_cast<uintptr_t>(bx.dy_out)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: unaligned y / dy_out", name);
    if (N * C * H * W == 0) return OFASR_OK;
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16)
        return launch_conv<f16_t, true, false, false, true>(name, da, f, dx, N, C, H, W, K, st, InputXf{}, StatOut{nullptr, 0},
                                                            BnFold{}, bx);
    return launch_conv<bf16_t, true, false, false, true>(name, da, f, dx, N, C, H, W, K, st, InputXf{}, StatOut{nullptr, 0},
                                                         BnFold{}, bx);
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT int ofasr_dwconv_fwd(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H,
                                  int64_t W, int K, int dtype, void* stream) {
    return conv_entry<false>("ofasr_dwconv_fwd", x, f, y, N, C, H, W, K, dtype, stream);
}

OFASR_EXPORT int ofasr_dwconv_dgrad(const void* dy, const float* f, void* dx, int64_t N, int64_t C, int64_t H,
                                    int64_t W, int K, int dtype, void* stream) {
    return conv_entry<true>("ofasr_dwconv_dgrad", dy, f, dx, N, C, H, W, K, dtype, stream);
}

OFASR_EXPORT size_t ofasr_dwconv_wgrad_workspace(int64_t N, int64_t C, int64_t H, int64_t W, int K) {
    (void)W;  // the bound below holds for every W
    if (N <= 0 || C <= 0 || K <= 0) return 0;
    const size_t strip = (size_t)wgrad_parts(N, C) * (size_t)C * (size_t)K * (size_t)K * sizeof(float);
    // the vector plan (chosen at launch when W and the pointers allow) keeps one partial per (plane, row chunk)
    // upper bound on slabs per plane for any W: gpw >= 1 and R >= min(32, H) => nslabs <= ceil(H / 4)
    const size_t vec = (size_t)N * (size_t)C * (size_t)cdiv(H > 0 ? H : 1, 4) * (size_t)K * (size_t)K * sizeof(float);
    return strip > vec ? strip : vec;
}

OFASR_EXPORT int ofasr_dwconv_wgrad(const void* dy, const void* x, float* df, int64_t N, int64_t C, int64_t H,
                                    int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    const char* name = "ofasr_dwconv_wgrad";
    int rc = check_conv_args(name, dy, x, df, N, C, H, W, K, dtype);
    if (rc) return rc;
    if (C == 0) return OFASR_OK;
    hipStream_t st = as_stream(stream);
    if (N * H * W == 0) {
        hipError_t e = hipMemsetAsync(df, 0, (size_t)C * K * K * sizeof(float), st);
        OFASR_REQUIRE(e == hipSuccess, OFASR_ERR_LAUNCH, "%s: memset failed", name);
        return OFASR_OK;
    }
    const size_t need = ofasr_dwconv_wgrad_workspace(N, C, H, W, K);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, need);
    float* ws = (float*)workspace;
    switch (dtype) {
        case OFASR_F32: return launch_wgrad<float>(name, dy, x, df, N, C, H, W, K, ws, st);
        case OFASR_F16: return launch_wgrad<f16_t>(name, dy, x, df, N, C, H, W, K, ws, st);
        default: return launch_wgrad<bf16_t>(name, dy, x, df, N, C, H, W, K, ws, st);
    }
}
