// pwconv.hip -- pointwise (1x1) convolution on in-place weight slices: fwd / dgrad / wgrad on the
// gfx950 matrix cores.
//
// Replaces DynamicPointConv2d.forward (reference ofa/elastic_nn/modules/dynamic_op.py:104-112:
// `weight[:out,:in].contiguous()` + F.conv2d 1x1) and its autograd.  Per image the op is the GEMM
//     Y_n[M x P] = W[M x K] . X_n[K x P]          (P = H*W pixels, contiguous in NCHW)
// fwd: (M,K) = (Cout,Cin); dgrad: (M,K) = (Cin,Cout) with W read transposed in place;
// wgrad: dW[M x K] = sum_n dY_n[M x P] . X_n[K x P]^T  (the reduction runs over pixels).
//
// Roofline: un-fused, the op moves B*(K+M)*P bytes per image for 2*M*K*P flops: 54 flop/B in bf16
// at (64 <-> 384), far left of the MFMA ridge (~310 flop/B), so the kernels are HBM-bound and are
// built around the memory system: every activation byte is read once and written once with
// 16-byte lanes, the weight slice is read in place (leading-dimension stride, no copy), and the
// MFMA work rides underneath.  fp32 activations use v_mfma_f32_32x32x2_f32 (exact fp32 fma
// chain, 1/16 of the bf16 rate) and are matrix-bound instead; that path exists for parity.
//
// Pixel <-> MFMA-column mapping.  With D[row = channel][col = pixel] the accumulator puts one
// pixel per lane, which would store 2 (bf16) or 4 bytes per lane.  The MFMA does not care which
// pixel a column is, so column c of sub-tile t is pixel P*c + t (P = 4 "fan-out", 2 "fan-in"):
// a lane then owns P adjacent pixels across its P accumulator tiles and stores them with one
// 8/16-byte access (256/512 contiguous bytes per half-wave).  For 16-bit inputs the matching
// B operand (8 channels of one pixel per lane) comes from an LDS image of the [channel][pixel]
// tile whose pixel positions are permuted at staging time, read with ds_read_b64_tr_b16.
//
// Two schedules cover the shapes of the MB block (and, slower, any other shape):
//   fan-out (K <= 64, any M)  expand fwd / project dgrad: one block = 128 pixels x 128 output rows; the
//                             64-channel X tile and the 128-row weight slab are staged once.
//   fan-in  (any K, M <= 64 per pass) project fwd / expand dgrad: K is walked in 64-channel chunks
//                             with the accumulators resident.
#include <stdlib.h>
#include "ofasr_common.h"

namespace ofasr {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int PW_THREADS = 256;
constexpr int PW_TILE = 128;            // pixels per block tile
constexpr int XROW16 = 320;             // bytes per channel row of a 16-bit X tile (256 + 64 pad)
constexpr int XROW32 = 512;             // bytes per channel row of an fp32 X tile
constexpr int WROW16 = 128;             // bytes per 64-k row of a 16-bit operand tile
constexpr int WROW32 = 256;             // bytes per 64-k row of an fp32 operand tile

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                       c, 0, 0, 0);
    }
};
template <> struct Mma16<f16_t> {
    static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                      0, 0, 0);
    }
};

template <typename T> struct Elem { static constexpr bool is16 = true; static constexpr int xrow = XROW16; static constexpr int wrow = WROW16; };
template <> struct Elem<float> { static constexpr bool is16 = false; static constexpr int xrow = XROW32; static constexpr int wrow = WROW32; };

// W(m, k) = w[m*sm + k*sk]: fwd reads the slice row-major, dgrad reads it transposed, both in place.
struct WView {
    const float* w;
    long long sm, sk;
    int M, K;
};

// ---- operand tile in LDS: [rows][64 k], 16-byte chunks XOR-swizzled so that 16 lanes reading the same
// chunk column of 16 different rows hit 16 different bank groups (ds_read_b128 conflict-free).
template <typename T>
__device__ __forceinline__ int wtile_off(int row, int k) {
    if constexpr (Elem<T>::is16) return row * WROW16 + ((((k >> 3) ^ ((row >> 1) & 7)) << 4)) + (k & 7) * 2;
    else return row * WROW32 + ((((k >> 2) ^ (row & 15)) << 4)) + (k & 3) * 4;
}
template <typename T>
__device__ __forceinline__ int wtile_chunk_off(int row, int chunk) {  // chunk = 16-byte unit of the row
    if constexpr (Elem<T>::is16) return row * WROW16 + ((chunk ^ ((row >> 1) & 7)) << 4);
    else return row * WROW32 + ((chunk ^ (row & 15)) << 4);
}

template <typename T>
__device__ __forceinline__ void lds_store_w(char* base, int off, float v) {
    if constexpr (Elem<T>::is16) *reinterpret_cast<uint16_t*>(base + off) = from_float<T>(v).v;
    else *reinterpret_cast<float*>(base + off) = v;
}

// stage W rows [m0, m0+ROWS) x k in [k0, k0+64) (zero outside the slice) into one operand tile.
// Vector form: every thread issues all of its 16-byte global loads first (ROWS/16 independent loads in
// flight), then converts and writes LDS -- the slice is L2-resident, so this is latency-, not
// bandwidth-limited and the loads must overlap.  Needs w 16-byte aligned and ldw % 4 == 0 (WVEC).
// A [ROWS x 64] weight chunk in registers (vector-loadable slices whose 4-element chunks are wholly inside or outside:
// K % 4 == 0 row-major, M % 4 == 0 transposed): wchunk_request issues all requests branch-free -- an outside chunk reads
// the slice's first chunk -- and wchunk_deposit writes them to the operand tile, zeroing the outside ones.  (A load under
// its bounds test ends the basic block with s_waitcnt vmcnt(0): NIT serial L2 round trips per tile.)
template <int ROWS> struct WChunk {
    static constexpr int NIT = ROWS * 16 / PW_THREADS;   // float4 chunks per thread
    float4 v[NIT];
    uint32_t ok;
};
__device__ __forceinline__ bool wchunk_whole(const WView& wv) { return wv.sk == 1 ? (wv.K % 4 == 0) : (wv.M % 4 == 0); }
template <int ROWS>
__device__ __forceinline__ void wchunk_request(WChunk<ROWS>& g, const WView& wv, int m0, int k0) {
    const int tid = threadIdx.x;
    g.ok = 0;
#pragma unroll
    for (int it = 0; it < WChunk<ROWS>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        long long off;
        bool ok;
        if (wv.sk == 1) {
            const int r = q >> 4, k = 4 * (q & 15);
            ok = m0 + r < wv.M && k0 + k < wv.K;
            off = (long long)(m0 + r) * wv.sm + k0 + k;
        } else {
            const int k = q / (ROWS / 4), r = 4 * (q - k * (ROWS / 4));
            ok = k0 + k < wv.K && m0 + r < wv.M;
            off = (long long)(k0 + k) * wv.sk + m0 + r;
        }
        g.v[it] = *reinterpret_cast<const float4*>(wv.w + (ok ? off : 0));
        g.ok |= (ok ? 1u : 0u) << it;
    }
}
template <typename T, int ROWS>
__device__ __forceinline__ void wchunk_deposit(char* Wt, const WChunk<ROWS>& g, const WView& wv) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < WChunk<ROWS>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        const float4 t = (g.ok >> it) & 1u ? g.v[it] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (wv.sk == 1) {
            const int r = q >> 4, k = 4 * (q & 15);
            if constexpr (Elem<T>::is16)
                *reinterpret_cast<uint2*>(Wt + wtile_off<T>(r, k)) = make_uint2(pack2<T>(t.x, t.y), pack2<T>(t.z, t.w));
            else
                *reinterpret_cast<float4*>(Wt + wtile_off<T>(r, k)) = t;
        } else {
            const int k = q / (ROWS / 4), r = 4 * (q - k * (ROWS / 4));
            lds_store_w<T>(Wt, wtile_off<T>(r, k), t.x);
            lds_store_w<T>(Wt, wtile_off<T>(r + 1, k), t.y);
            lds_store_w<T>(Wt, wtile_off<T>(r + 2, k), t.z);
            lds_store_w<T>(Wt, wtile_off<T>(r + 3, k), t.w);
        }
    }
}

template <typename T, int ROWS, bool WVEC>
__device__ __forceinline__ void stage_w_tile(char* Wt, const WView& wv, int m0, int k0) {
    const int tid = threadIdx.x;
    if constexpr (!WVEC) {
        if (wv.sk == 1) {
            for (int e = tid; e < ROWS * 64; e += PW_THREADS) {
                const int r = e >> 6, k = e & 63;
                const int m = m0 + r, kk = k0 + k;
                const float v = (m < wv.M && kk < wv.K) ? wv.w[(long long)m * wv.sm + kk] : 0.f;
                lds_store_w<T>(Wt, wtile_off<T>(r, k), v);
            }
        } else {
            for (int e = tid; e < ROWS * 64; e += PW_THREADS) {
                const int k = e / ROWS, r = e - k * ROWS;
                const int m = m0 + r, kk = k0 + k;
                const float v = (m < wv.M && kk < wv.K) ? wv.w[(long long)m * wv.sm + (long long)kk * wv.sk] : 0.f;
                lds_store_w<T>(Wt, wtile_off<T>(r, k), v);
            }
        }
    } else {
        constexpr int NIT = ROWS * 16 / PW_THREADS;  // float4 chunks per thread
        float4 v[NIT];
        if (wchunk_whole(wv)) {   // uniform
            WChunk<ROWS> g;
            wchunk_request<ROWS>(g, wv, m0, k0);
            wchunk_deposit<T, ROWS>(Wt, g, wv);
            return;
        }
        if (wv.sk == 1) {  // row-major slice: chunk = 4 consecutive k of one row
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * PW_THREADS;
                const int r = q >> 4, k = 4 * (q & 15);
                const int m = m0 + r, kk = k0 + k;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < wv.M) {
                    const float* src = wv.w + (long long)m * wv.sm + kk;
                    if (kk + 3 < wv.K) v[it] = *reinterpret_cast<const float4*>(src);
                    else {
                        if (kk < wv.K) v[it].x = src[0];
                        if (kk + 1 < wv.K) v[it].y = src[1];
                        if (kk + 2 < wv.K) v[it].z = src[2];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * PW_THREADS;
                const int r = q >> 4, k = 4 * (q & 15);
                if constexpr (Elem<T>::is16)
                    *reinterpret_cast<uint2*>(Wt + wtile_off<T>(r, k)) =
                        make_uint2(pack2<T>(v[it].x, v[it].y), pack2<T>(v[it].z, v[it].w));
                else
                    *reinterpret_cast<float4*>(Wt + wtile_off<T>(r, k)) = v[it];
            }
        } else {  // transposed view (dgrad): chunk = 4 consecutive rows m of one k
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * PW_THREADS;
                const int k = q / (ROWS / 4), r = 4 * (q - k * (ROWS / 4));
                const int m = m0 + r, kk = k0 + k;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kk < wv.K) {
                    const float* src = wv.w + (long long)kk * wv.sk + m;
                    if (m + 3 < wv.M) v[it] = *reinterpret_cast<const float4*>(src);
                    else {
                        if (m < wv.M) v[it].x = src[0];
                        if (m + 1 < wv.M) v[it].y = src[1];
                        if (m + 2 < wv.M) v[it].z = src[2];
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int q = tid + it * PW_THREADS;
                const int k = q / (ROWS / 4), r = 4 * (q - k * (ROWS / 4));
                lds_store_w<T>(Wt, wtile_off<T>(r, k), v[it].x);
                lds_store_w<T>(Wt, wtile_off<T>(r + 1, k), v[it].y);
                lds_store_w<T>(Wt, wtile_off<T>(r + 2, k), v[it].z);
                lds_store_w<T>(Wt, wtile_off<T>(r + 3, k), v[it].w);
            }
        }
    }
}

// ---- X tile staging: channels [k0, k0+64) x pixels [p0, p0+128) of image plane block xn ([K][HW]).
// PX = pixels per lane (4 fan-out, 2 fan-in).  16-bit position of pixel q in [0,128):
//   PX=4: 32*(q&3) + (q>>2)            PX=2: 64*(q>>6) + 32*(q&1) + ((q&63)>>1)
template <int PX>
__device__ __forceinline__ int pos16(int q) {
    return PX == 4 ? (32 * (q & 3) + (q >> 2)) : (64 * (q >> 6) + 32 * (q & 1) + ((q & 63) >> 1));
}

// fused BN + ReLU6 of 8 packed 16-bit values of channel ch (InputXf, ofasr_common.h)
template <typename T>
__device__ __forceinline__ uint4 xf_apply8_core(uint4 v, float sc, float mu, float b);
template <typename T>
__device__ __forceinline__ uint4 xf_apply8(uint4 v, const InputXf& xf, int ch) {
    const float sc = xf.scale[ch], mu = xf.mean[ch];
    return xf_apply8_core<T>(v, sc, mu, fmaf(mu, sc, xf.shift[ch]));
}
template <typename T>
__device__ __forceinline__ uint4 xf_apply8_core(uint4 v, float sc, float mu, float b) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T lo, hi;
        lo.v = (uint16_t)(w[i] & 0xffffu);
        hi.v = (uint16_t)(w[i] >> 16);
        const float a = fminf(fmaxf(fmaf(to_float(lo) - mu, sc, b), 0.f), 6.f);
        const float c = fminf(fmaxf(fmaf(to_float(hi) - mu, sc, b), 0.f), 6.f);
        w[i] = pack2<T>(a, c);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// BN(+ReLU6) backward of 8 packed gradient values `g` of one channel given the 8 pre-BN values `yv` (BwdXf)
template <typename T>
__device__ __forceinline__ uint4 bx_apply8(uint4 g, uint4 yv, float mu, float sc, float xb, float ka, float kbi) {
    uint32_t w[4] = {g.x, g.y, g.z, g.w};
    const uint32_t yw[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        T lo, hi, ylo, yhi;
        lo.v = (uint16_t)(w[i] & 0xffffu);
        hi.v = (uint16_t)(w[i] >> 16);
        ylo.v = (uint16_t)(yw[i] & 0xffffu);
        yhi.v = (uint16_t)(yw[i] >> 16);
        const float t0 = to_float(ylo) - mu, t1 = to_float(yhi) - mu;
        const float p0 = fmaf(t0, sc, xb), p1 = fmaf(t1, sc, xb);
        const float z0 = (p0 > 0.f && p0 < 6.f) ? to_float(lo) : 0.f;
        const float z1 = (p1 > 0.f && p1 < 6.f) ? to_float(hi) : 0.f;
        w[i] = pack2<T>(fmaf(-t0, kbi, fmaf(sc, z0, -ka)), fmaf(-t1, kbi, fmaf(sc, z1, -ka)));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// A [64 channels x 128 pixels] X chunk in registers (aligned tensors: 16-byte vectors wholly inside or outside): request
// = all vectors issued branch-free (a vector outside the tensor reads the image's first one), deposit = fused BN + ReLU6
// where asked for, zero the outside ones, write the staging layout of stage_x_tile.
template <typename T> struct XChunk {
    static constexpr int NIT = Elem<T>::is16 ? 4 : 8;     // 16-byte vectors per thread
    static constexpr int VPR = Elem<T>::is16 ? 16 : 32;   // vectors per channel row
    static constexpr int EPV = Elem<T>::is16 ? 8 : 4;     // elements per vector
    uint4 v[NIT];
    uint32_t ok;
};
template <typename T>
__device__ __forceinline__ void xchunk_request(XChunk<T>& g, const T* __restrict__ xn, int K, int HW, int k0, int p0) {
    const int tid = threadIdx.x;
    g.ok = 0;
#pragma unroll
    for (int it = 0; it < XChunk<T>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        const int k = q / XChunk<T>::VPR, m = q % XChunk<T>::VPR;
        const int px = p0 + XChunk<T>::EPV * m;
        const bool ok = k0 + k < K && px < HW;
        g.v[it] = *reinterpret_cast<const uint4*>(xn + (ok ? (long long)(k0 + k) * HW + px : 0));
        g.ok |= (ok ? 1u : 0u) << it;
    }
}
template <typename T, int PX, bool XF>
__device__ __forceinline__ void xchunk_deposit(char* Xs, const XChunk<T>& g, int K, int k0, InputXf xf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < XChunk<T>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        const int k = q / XChunk<T>::VPR, m = q % XChunk<T>::VPR;
        uint4 v = g.v[it];
        if constexpr (Elem<T>::is16 && XF) v = xf_apply8<T>(v, xf, k0 + k < K ? k0 + k : K - 1);
        const uint32_t msk = (g.ok >> it) & 1u ? 0xffffffffu : 0u;
        v = make_uint4(v.x & msk, v.y & msk, v.z & msk, v.w & msk);
        if constexpr (Elem<T>::is16) {
            char* row = Xs + k * XROW16;
            if (PX == 4) {
                // pixel 8m+i -> position 32*(i&3) + 2m + (i>>2): pairs (i, i+4) are adjacent
                uint32_t* r32 = reinterpret_cast<uint32_t*>(row);
                r32[(32 * 0 + 2 * m) >> 1] = (v.x & 0xffffu) | (v.z << 16);
                r32[(32 * 1 + 2 * m) >> 1] = (v.x >> 16) | (v.z & 0xffff0000u);
                r32[(32 * 2 + 2 * m) >> 1] = (v.y & 0xffffu) | (v.w << 16);
                r32[(32 * 3 + 2 * m) >> 1] = (v.y >> 16) | (v.w & 0xffff0000u);
            } else {
                // pixel 8m+i -> position 64*(m>>3) + 32*(i&1) + 4*(m&7) + (i>>1)
                const int base = 64 * (m >> 3) + 4 * (m & 7);
                const uint2 ev = make_uint2((v.x & 0xffffu) | (v.y << 16), (v.z & 0xffffu) | (v.w << 16));
                const uint2 od = make_uint2((v.x >> 16) | (v.y & 0xffff0000u), (v.z >> 16) | (v.w & 0xffff0000u));
                *reinterpret_cast<uint2*>(row + (base)*2) = ev;
                *reinterpret_cast<uint2*>(row + (base + 32) * 2) = od;
            }
        } else {
            *reinterpret_cast<uint4*>(Xs + k * XROW32 + m * 16) = v;
        }
    }
}

template <typename T, int PX, bool ALIGNED, bool XF = false>
__device__ __forceinline__ void stage_x_tile(char* Xs, const T* __restrict__ xn, int K, int HW, int k0, int p0,
                                             InputXf xf = InputXf{}) {
    const int tid = threadIdx.x;
    if constexpr (Elem<T>::is16) {
        if (ALIGNED) {
            // 64 rows x 16 vectors of 8 pixels (16 B); HW % 8 == 0 so a vector is all-in or all-out
            XChunk<T> g;
            xchunk_request<T>(g, xn, K, HW, k0, p0);
            xchunk_deposit<T, PX, XF>(Xs, g, K, k0, xf);
        } else {
            for (int e = tid; e < 64 * PW_TILE; e += PW_THREADS) {
                const int k = e >> 7, q = e & 127;
                const int px = p0 + q;
                uint16_t v = 0;
                if (k0 + k < K && px < HW) v = reinterpret_cast<const uint16_t*>(xn)[(long long)(k0 + k) * HW + px];
                *reinterpret_cast<uint16_t*>(Xs + k * XROW16 + pos16<PX>(q) * 2) = v;
            }
        }
    } else {
        if (ALIGNED) {
            // 64 rows x 32 vectors of 4 pixels (16 B); HW % 4 == 0
            XChunk<T> g;
            xchunk_request<T>(g, xn, K, HW, k0, p0);
            xchunk_deposit<T, PX, XF>(Xs, g, K, k0, xf);
        } else {
            for (int e = tid; e < 64 * PW_TILE; e += PW_THREADS) {
                const int k = e >> 7, q = e & 127;
                const int px = p0 + q;
                float v = 0.f;
                if (k0 + k < K && px < HW) v = reinterpret_cast<const float*>(xn)[(long long)(k0 + k) * HW + px];
                *reinterpret_cast<float*>(Xs + k * XROW32 + q * 4) = v;
            }
        }
    }
}

// B fragment (8 channels of one pixel-column per lane) of 32-column sub-tile starting at LDS position
// `pos0`, k-step s (channels 16s..16s+15), by two transposing reads.
__device__ __forceinline__ s16x8 read_b_frag16(const char* Xs, int pos0, int s, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kb = 16 * s + 8 * (g >> 1);
    const int colb = (pos0 + 16 * (g & 1) + 4 * pp) * 2;
    const lds_s16x4* p0 = (const lds_s16x4*)(Xs + (kb + q) * XROW16 + colb);
    const lds_s16x4* p1 = (const lds_s16x4*)(Xs + (kb + 4 + q) * XROW16 + colb);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row of accumulator register `reg` for lane half h (C/D layout of the 32x32 MFMA)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// store PX adjacent pixels of one output row
template <typename T, int PX, bool ALIGNED>
__device__ __forceinline__ void store_px(T* __restrict__ dst, int px, int HW, const float* v) {
    if (ALIGNED) {
        if (px < HW) {
            if constexpr (Elem<T>::is16) {
                if (PX == 4) *reinterpret_cast<uint2*>(dst + px) = make_uint2(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]));
                else *reinterpret_cast<uint32_t*>(dst + px) = pack2<T>(v[0], v[1]);
            } else {
                if (PX == 4) *reinterpret_cast<float4*>(dst + px) = make_float4(v[0], v[1], v[2], v[3]);
                else *reinterpret_cast<float2*>(dst + px) = make_float2(v[0], v[1]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < PX; ++t)
            if (px + t < HW) dst[px + t] = from_float<T>(v[t]);
    }
}

// v[t] += addend[px + t] (the residual / shortcut gradient accumulated in the epilogue, one rounding on store)
template <typename T, int PX, bool ALIGNED>
__device__ __forceinline__ void add_px(const T* __restrict__ src, int px, int HW, float* v) {
    if (ALIGNED) {
        if (px < HW) {
            if constexpr (Elem<T>::is16) {
                uint32_t w[PX / 2];
                if (PX == 4) {
                    const uint2 u = *reinterpret_cast<const uint2*>(src + px);
                    w[0] = u.x;
                    w[PX / 2 - 1] = u.y;
                } else {
                    w[0] = *reinterpret_cast<const uint32_t*>(src + px);
                }
#pragma unroll
                for (int i = 0; i < PX / 2; ++i) {
                    T lo, hi;
                    lo.v = (uint16_t)(w[i] & 0xffffu);
                    hi.v = (uint16_t)(w[i] >> 16);
                    v[2 * i] += to_float(lo);
                    v[2 * i + 1] += to_float(hi);
                }
            } else {
#pragma unroll
                for (int t = 0; t < PX; ++t) v[t] += to_float(src[px + t]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < PX; ++t)
            if (px + t < HW) v[t] += to_float(src[px + t]);
    }
}

// (half_wave_row_sums / half_wave_row_reg: ofasr_common.h)

// statistics of NSUB accumulator sub-tiles (PX = NSUB adjacent pixels per lane starting at px) of one 32-row block
template <typename T, int NSUB>
__device__ __forceinline__ void stat_epilogue(const f32x16* acc, int px, int HW, int c, int h, int row_base, int mloc,
                                              long long chan_base, StatOut so, int unit) {
    float s[16], q[16];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        float ss = 0.f, qq = 0.f;
        float vs[NSUB];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) vs[t] = acc[t][reg];
        if constexpr (Elem<T>::is16) {   // the values as stored: rounded through the very pair conversions the store issues
            static_assert(NSUB % 2 == 0, "pairs");
#pragma unroll
            for (int t = 0; t < NSUB; t += 2) unpack2<T>(pack2<T>(vs[t], vs[t + 1]), vs[t], vs[t + 1]);
        }
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            const float v = vs[t];
            if (px + t < HW) {
                ss += v;
                qq = fmaf(v, v, qq);
            }
        }
        s[reg] = ss;
        q[reg] = qq;
    }
    half_wave_row_sums(s, q, c);
    const int r = row_base + acc_row(half_wave_row_reg(c), h);
    if ((c & 1) == 0 && r < mloc) so.partial[(chan_base + r) * so.P + unit] = make_float2(s[0], q[0]);
}

// ------------------------------------------------------------------------------------ fan-out
// K <= 64.  One block = one 128-pixel tile x one 128-row slab of outputs (grid.y); wave w owns output
// row block w of the slab: A (weights) from the swizzled LDS operand tile, B (pixels) by transposing
// reads of the X tile, 16 MFMAs, packed 8/16-byte stores.  ~36 KB LDS and ~100 VGPRs per block keep
// 4 blocks (16 waves) per CU so loads of one block overlap the MFMA/store phase of the others.
constexpr int FO_ROWS = 128;
template <typename T, bool ALIGNED, bool WVEC, bool XF = false>
__global__ void __launch_bounds__(PW_THREADS) pw_fanout_kernel(const T* __restrict__ x, WView wv, T* __restrict__ y,
                                                               int HW, int tiles_per_img, InputXf xf = InputXf{},
                                                               const T* __restrict__ addend = nullptr,
                                                               StatOut so = StatOut{nullptr, 0}) {
    __shared__ __attribute__((aligned(16))) char Ws[FO_ROWS * Elem<T>::wrow];
    __shared__ __attribute__((aligned(16))) char Xs[64 * Elem<T>::xrow];
    const int m_base = blockIdx.y * FO_ROWS;
    const int mloc = min(FO_ROWS, wv.M - m_base);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    const int n = tile / tiles_per_img;
    const int p0 = (tile - n * tiles_per_img) * PW_TILE;
    const T* xn = x + (long long)n * wv.K * HW;
    T* yn = y + ((long long)n * wv.M + m_base) * HW;

    if constexpr (Elem<T>::is16 && ALIGNED && WVEC) {
        // one latency phase: the 8 weight loads (L2) and the 4 X loads (HBM) of a thread are all requested before either
        // tile is written to LDS (staging X and then W costs an HBM round trip plus an L2 round trip)
        const int tid = threadIdx.x;
        const bool rowmajor = wv.sk == 1;
        float4 wr[8];
        // WVEC implies K % 4 == 0 and M % 4 == 0 (launch_gemm): every 16-byte chunk is wholly inside or outside the slice -> branch-free requests
        // (clamped address, zeroed when written to LDS); a conditional load ends its basic block with s_waitcnt
        // vmcnt(0), which turns the requests below into serial round trips
        uint32_t okbits = 0;
        {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q = tid + it * PW_THREADS;
                long long off;
                bool ok;
                if (rowmajor) {
                    const int r = q >> 4, kk = 4 * (q & 15);
                    ok = r < mloc && kk < wv.K;
                    off = (long long)(m_base + r) * wv.sm + kk;
                } else {
                    const int k = q >> 5, r = 4 * (q & 31);
                    ok = k < wv.K && m_base + r < wv.M;
                    off = (long long)k * wv.sk + m_base + r;
                }
                wr[it] = *reinterpret_cast<const float4*>(wv.w + (ok ? off : 0));
                okbits |= (ok ? 1u : 0u) << it;
            }
        }
        uint4 xr[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {   // HW % 8 == 0 (ALIGNED): a vector is wholly inside or outside the plane
            const int q = tid + it * PW_THREADS;
            const int k = q >> 4, px = p0 + 8 * (q & 15);
            const bool ok = k < wv.K && px < HW;
            xr[it] = *reinterpret_cast<const uint4*>(xn + (ok ? (long long)k * HW + px : 0));
            okbits |= (ok ? 256u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (!((okbits >> (8 + it)) & 1u)) xr[it] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if (!((okbits >> it) & 1u)) wr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = tid + it * PW_THREADS;
            if (rowmajor) {
                const int r = q >> 4, k = 4 * (q & 15);
                *reinterpret_cast<uint2*>(Ws + wtile_off<T>(r, k)) =
                    make_uint2(pack2<T>(wr[it].x, wr[it].y), pack2<T>(wr[it].z, wr[it].w));
            } else {
                const int k = q >> 5, r = 4 * (q & 31);
                lds_store_w<T>(Ws, wtile_off<T>(r, k), wr[it].x);
                lds_store_w<T>(Ws, wtile_off<T>(r + 1, k), wr[it].y);
                lds_store_w<T>(Ws, wtile_off<T>(r + 2, k), wr[it].z);
                lds_store_w<T>(Ws, wtile_off<T>(r + 3, k), wr[it].w);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            const int k = q >> 4, m = q & 15;
            uint4 v = xr[it];
            if constexpr (XF) {
                if (k < wv.K && p0 + 8 * m < HW) v = xf_apply8<T>(v, xf, k);
            }
            // pixel 8m+i -> position 32*(i&3) + 2m + (i>>2): pairs (i, i+4) are adjacent   (stage_x_tile, PX = 4)
            uint32_t* r32 = reinterpret_cast<uint32_t*>(Xs + k * XROW16);
            r32[(32 * 0 + 2 * m) >> 1] = (v.x & 0xffffu) | (v.z << 16);
            r32[(32 * 1 + 2 * m) >> 1] = (v.x >> 16) | (v.z & 0xffff0000u);
            r32[(32 * 2 + 2 * m) >> 1] = (v.y & 0xffffu) | (v.w << 16);
            r32[(32 * 3 + 2 * m) >> 1] = (v.y >> 16) | (v.w & 0xffff0000u);
        }
    } else {
        stage_x_tile<T, 4, ALIGNED, XF>(Xs, xn, wv.K, HW, 0, p0, xf);
        stage_w_tile<T, FO_ROWS, WVEC>(Ws, wv, m_base, 0);
    }
    __syncthreads();

    const int cb = wave;
    if (32 * cb >= mloc) return;
    const int row = 32 * cb + c;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = zero16();
    if constexpr (Elem<T>::is16) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (16 * s < wv.K) {
                const s16x8 af = *reinterpret_cast<const s16x8*>(Ws + wtile_chunk_off<T>(row, 2 * s + h));
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = Mma16<T>::run(af, read_b_frag16(Xs, 32 * t, s, lane), acc[t]);
            }
        }
    } else {
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const float4 a4 = *reinterpret_cast<const float4*>(Ws + wtile_chunk_off<T>(row, 8 * h + s4));
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int k = 32 * h + 4 * s4 + ss;  // MFMA k-slot h of step s <-> channel 32h + s
                const float4 b4 = *reinterpret_cast<const float4*>(Xs + k * XROW32 + c * 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b4.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b4.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b4.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b4.w, acc[3], 0, 0, 0);
            }
        }
    }
    if constexpr (Elem<T>::is16 && ALIGNED) {
        // a wave whose 32 rows x 128 pixels lie inside the tensor writes them as straight-line code (no store or addend
        // load under a lane- or row-dependent branch: each would be a serial round trip)
        if (32 * cb + 32 <= mloc && p0 + PW_TILE <= HW && addend == nullptr) {
            const int px = p0 + 4 * c;
            T* yw = yn + (long long)(32 * cb) * HW + px;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                *reinterpret_cast<uint2*>(yw + (long long)acc_row(reg, h) * HW) =
                    make_uint2(pack2<T>(acc[0][reg], acc[1][reg]), pack2<T>(acc[2][reg], acc[3][reg]));
            if (so.partial) stat_epilogue<T, 4>(acc, px, HW, c, h, 32 * cb, mloc, m_base, so, tile);
            return;
        }
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = 32 * cb + acc_row(reg, h);
        if (r < mloc) {
            float v[4] = {acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]};
            if (addend)
                add_px<T, 4, ALIGNED>(addend + ((long long)n * wv.M + m_base + r) * HW, p0 + 4 * c, HW, v);
            store_px<T, 4, ALIGNED>(yn + (long long)r * HW, p0 + 4 * c, HW, v);
        }
    }
    if (so.partial) stat_epilogue<T, 4>(acc, p0 + 4 * c, HW, c, h, 32 * cb, mloc, m_base, so, tile);
}


// ------------------------------------------------------------------------------ fan-out, slab walk
// Same tile maths as pw_fanout_kernel for the hot shape class (16-bit, aligned, vector weights, K == 64, M == NSLAB*128,
// HW % 128 == 0): ONE block per 128-pixel tile walks the NSLAB 128-row output slabs.  The X tile is staged once (the
// per-slab grid re-reads it from L2 for every slab) and the weight slab of round s+1 is requested before the MFMAs of
// round s.  The kernel is store-dominated (a 64 -> 384 expand writes 6x what it reads), so what matters is that a wave
// never waits for its own stores: vector-memory operations retire in order, so (a) the rounds are fully unrolled
// straight-line code without a single divergent or wave-dependent branch -- then the compiler's s_waitcnt for the
// prefetched weights is exact (vmcnt = the 16-17 younger stores) instead of vmcnt(0) -- and (b) the barriers order LDS
// only (__syncthreads() would drain the stores too).
// workgroup barrier that orders LDS traffic only
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ void fo_load_w(float4 (&wr)[8], const WView& wv, bool rowmajor, int m_base, int tid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = tid + it * PW_THREADS;
        const long long off = rowmajor ? (long long)(m_base + (q >> 4)) * wv.sm + 4 * (q & 15)
                                       : (long long)(q >> 5) * wv.sk + m_base + 4 * (q & 31);
        wr[it] = *reinterpret_cast<const float4*>(wv.w + off);
    }
}
template <typename T>
__device__ __forceinline__ void fo_store_w(char* Ws, const float4 (&wr)[8], bool rowmajor, int tid) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int q = tid + it * PW_THREADS;
        if (rowmajor) {
            const int r = q >> 4, k = 4 * (q & 15);
            *reinterpret_cast<uint2*>(Ws + wtile_off<T>(r, k)) =
                make_uint2(pack2<T>(wr[it].x, wr[it].y), pack2<T>(wr[it].z, wr[it].w));
        } else {
            const int k = q >> 5, r = 4 * (q & 31);
            lds_store_w<T>(Ws, wtile_off<T>(r, k), wr[it].x);
            lds_store_w<T>(Ws, wtile_off<T>(r + 1, k), wr[it].y);
            lds_store_w<T>(Ws, wtile_off<T>(r + 2, k), wr[it].z);
            lds_store_w<T>(Ws, wtile_off<T>(r + 3, k), wr[it].w);
        }
    }
}

// STAT: 0 none; 1 BatchNorm statistics of the rows written (forward: StatOut); 2 BatchNorm-backward sums of the rows
// written (the kernel then produces the gradient da of an activated tensor: BwdStatOut) -- the y values of the
// positions a lane writes are requested before the round's MFMAs, 16 x 8 bytes mirroring its 16 stores.
template <typename T, int NSLAB, int STAT>
__global__ void __launch_bounds__(PW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
pw_fanout_slabs_kernel(const T* __restrict__ x, WView wv, T* __restrict__ y, int HW, int tiles_per_img, StatOut so,
                       BwdStatOut bs) {
    static_assert(Elem<T>::is16, "16-bit activations only");
    __shared__ __attribute__((aligned(16))) char Ws[FO_ROWS * Elem<T>::wrow];
    __shared__ __attribute__((aligned(16))) char Xs[64 * Elem<T>::xrow];
    __shared__ float4 btab[STAT == 2 ? NSLAB * FO_ROWS : 1];   // per output row: mean, scale, beta = shift + mean*scale
    if constexpr (STAT == 2) {
        for (int i = threadIdx.x; i < NSLAB * FO_ROWS; i += PW_THREADS) {
            const float mu = bs.mean[i], sc = bs.scale[i];
            btab[i] = make_float4(mu, sc, fmaf(mu, sc, bs.shift[i]), 0.f);
        }
    }
    const int lane = lane_id();
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    const int n = tile / tiles_per_img;
    const int p0 = (tile - n * tiles_per_img) * PW_TILE;
    const T* xn = x + (long long)n * wv.K * HW;
    const bool rowmajor = wv.sk == 1;
    float4 wr[8];
    fo_load_w<T>(wr, wv, rowmajor, 0, tid);
    {
        uint4 xr[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            xr[it] = *reinterpret_cast<const uint4*>(xn + (long long)(q >> 4) * HW + p0 + 8 * (q & 15));
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            const int k = q >> 4, m = q & 15;
            const uint4 v = xr[it];
            // pixel 8m+i -> position 32*(i&3) + 2m + (i>>2): pairs (i, i+4) are adjacent   (stage_x_tile, PX = 4)
            uint32_t* r32 = reinterpret_cast<uint32_t*>(Xs + k * XROW16);
            r32[(32 * 0 + 2 * m) >> 1] = (v.x & 0xffffu) | (v.z << 16);
            r32[(32 * 1 + 2 * m) >> 1] = (v.x >> 16) | (v.z & 0xffff0000u);
            r32[(32 * 2 + 2 * m) >> 1] = (v.y & 0xffffu) | (v.w << 16);
            r32[(32 * 3 + 2 * m) >> 1] = (v.y >> 16) | (v.w & 0xffff0000u);
        }
    }
    const int cb = wave;
    const int row = 32 * cb + c;
    const int px = p0 + 4 * c;
#pragma unroll
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int m_base = sl * FO_ROWS;
        fo_store_w<T>(Ws, wr, rowmajor, tid);
        lds_barrier();                                     // Ws (and, in round 0, Xs) visible
        if (sl + 1 < NSLAB) fo_load_w<T>(wr, wv, rowmajor, m_base + FO_ROWS, tid);   // in flight over this round
        uint2 yr[STAT == 2 ? 16 : 1];
        if constexpr (STAT == 2) {
            const T* ysrc = reinterpret_cast<const T*>(bs.y) + ((long long)n * wv.M + m_base + 32 * cb) * HW + px;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) yr[reg] = *reinterpret_cast<const uint2*>(ysrc + (long long)acc_row(reg, h) * HW);
        }
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const s16x8 af = *reinterpret_cast<const s16x8*>(Ws + wtile_chunk_off<T>(row, 2 * s + h));
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = Mma16<T>::run(af, read_b_frag16(Xs, 32 * t, s, lane), acc[t]);
        }
        T* yn = y + ((long long)n * wv.M + m_base + 32 * cb) * HW + px;
        if constexpr (STAT == 2) {
            // sums over this lane's 4 pixels per row of dz = (0 < BN(y) < 6) ? da : 0 and dz * (y - mean), da as stored
            float sv[16], qv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t w01 = pack2<T>(acc[0][reg], acc[1][reg]), w23 = pack2<T>(acc[2][reg], acc[3][reg]);
                *reinterpret_cast<uint2*>(yn + (long long)acc_row(reg, h) * HW) = make_uint2(w01, w23);
                const float4 cst = btab[m_base + 32 * cb + acc_row(reg, h)];
                float da[4], yv[4];
                unpack2<T>(w01, da[0], da[1]);
                unpack2<T>(w23, da[2], da[3]);
                unpack2<T>(yr[reg].x, yv[0], yv[1]);
                unpack2<T>(yr[reg].y, yv[2], yv[3]);
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float tt = yv[t] - cst.x;
                    const float pre = fmaf(tt, cst.y, cst.z);
                    const float dz = (pre > 0.f && pre < 6.f) ? da[t] : 0.f;
                    ss += dz;
                    qq = fmaf(dz, tt, qq);
                }
                sv[reg] = ss;
                qv[reg] = qq;
            }
            half_wave_row_sums(sv, qv, c);
            const int r = m_base + 32 * cb + acc_row(half_wave_row_reg(c), h);
            bs.partial[(long long)r * bs.P + tile] = make_float2(sv[0], qv[0]);
        } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            *reinterpret_cast<uint2*>(yn + (long long)acc_row(reg, h) * HW) =
                make_uint2(pack2<T>(acc[0][reg], acc[1][reg]), pack2<T>(acc[2][reg], acc[3][reg]));
        }
        if constexpr (STAT == 1) {
            float sv[16], qv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                float ss = 0.f, qq = 0.f;
                float vs[4];   // the values as stored: unpacked from the very pair conversions the store above issued
                unpack2<T>(pack2<T>(acc[0][reg], acc[1][reg]), vs[0], vs[1]);
                unpack2<T>(pack2<T>(acc[2][reg], acc[3][reg]), vs[2], vs[3]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ss += vs[t];
                    qq = fmaf(vs[t], vs[t], qq);
                }
                sv[reg] = ss;
                qv[reg] = qq;
            }
            half_wave_row_sums(sv, qv, c);
            // the two lanes of a b0 pair hold the same totals and store them to the same place (no lane-dependent branch)
            const int r = m_base + 32 * cb + acc_row(half_wave_row_reg(c), h);
            so.partial[(long long)r * so.P + tile] = make_float2(sv[0], qv[0]);
        }
        if (sl + 1 < NSLAB) lds_barrier();                 // every wave is done reading Ws
    }
}

// ------------------------------------------------------------------------------------- fan-in
// any K, 64 output rows per grid.y pass.  K is walked in 64-channel chunks: each chunk stages its X
// tile (HBM) and its [64 x 64] weight chunk (L2) and adds into accumulators that stay in registers.
// wave w: output row block cb = w&1, pixel half hh = w>>1 (64 pixels, 2 per lane).
// Measured timeline of one launch at 16x384->64x64x64 (wall_clock64 per block, all 512 blocks resident and in
// lockstep): the X loads of a block take ~10 us to arrive (the tile pattern alone streams at 5.7 TB/s,
// tools/probe/tile_read_probe.hip), vector-memory returns are in order so L2 weight loads issued after HBM X loads
// wait behind them, the 6 LDS/MFMA rounds take 3.4 us and the stores 1.3 us -- and nothing overlaps because there is
// a single generation of blocks.  Requesting every X chunk up front into registers (+ the whole weight slab in LDS)
// made the isolated launch 10 % faster but the step slower (68 KB LDS, 2 blocks/CU); the next step is a persistent
// block that streams tiles through a double buffer.
template <typename T, bool ALIGNED, bool WVEC, bool XF = false>
__global__ void __launch_bounds__(PW_THREADS) pw_fanin_kernel(const T* __restrict__ x, WView wv, T* __restrict__ y,
                                                              int HW, int tiles_per_img, int kchunks,
                                                              InputXf xf = InputXf{},
                                                              const T* __restrict__ addend = nullptr,
                                                              StatOut so = StatOut{nullptr, 0}) {
    __shared__ __attribute__((aligned(16))) char Ws[64 * Elem<T>::wrow];
    __shared__ __attribute__((aligned(16))) char Xs[64 * Elem<T>::xrow];
    const int m_base = blockIdx.y * 64;
    const int mloc = min(64, wv.M - m_base);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int cb = wave & 1, hh = wave >> 1;
    const int tile = blockIdx.x;
    const int n = tile / tiles_per_img;
    const int p0 = (tile - n * tiles_per_img) * PW_TILE;
    const T* xn = x + (long long)n * wv.K * HW;
    T* yn = y + ((long long)n * wv.M + m_base) * HW;
    f32x16 acc[2];
    acc[0] = zero16();
    acc[1] = zero16();
    const int row = 32 * cb + c;
    // aligned tensors and vector-loadable weights: chunk kc + 1 is requested into registers before the MFMAs of chunk kc
    // (as pw_fanin_pipe_kernel, which serves the 16-bit hot shapes; this is what the fp32 1x1 forward / input gradient
    // run on): its round trip runs under the round instead of after it
    const bool piped = ALIGNED && WVEC && wchunk_whole(wv);   // uniform
    XChunk<T> xg;
    WChunk<64> wg;
    if (piped) {
        xchunk_request<T>(xg, xn, wv.K, HW, 0, p0);
        wchunk_request<64>(wg, wv, m_base, 0);
    }
    for (int kc = 0; kc < kchunks; ++kc) {
        if (kc) __syncthreads();
        if (piped) {
            xchunk_deposit<T, 2, XF>(Xs, xg, wv.K, kc * 64, xf);
            wchunk_deposit<T, 64>(Ws, wg, wv);
        } else {
            stage_x_tile<T, 2, ALIGNED, XF>(Xs, xn, wv.K, HW, kc * 64, p0, xf);
            stage_w_tile<T, 64, WVEC>(Ws, wv, m_base, kc * 64);
        }
        __syncthreads();
        if (piped && kc + 1 < kchunks) {
            xchunk_request<T>(xg, xn, wv.K, HW, (kc + 1) * 64, p0);
            wchunk_request<64>(wg, wv, m_base, (kc + 1) * 64);
        }
        if constexpr (Elem<T>::is16) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const s16x8 af = *reinterpret_cast<const s16x8*>(Ws + wtile_chunk_off<T>(row, 2 * s + h));
                const s16x8 b0 = read_b_frag16(Xs, 64 * hh, s, lane);
                const s16x8 b1 = read_b_frag16(Xs, 64 * hh + 32, s, lane);
                acc[0] = Mma16<T>::run(af, b0, acc[0]);
                acc[1] = Mma16<T>::run(af, b1, acc[1]);
            }
        } else {
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const float4 a4 = *reinterpret_cast<const float4*>(Ws + wtile_chunk_off<T>(row, 8 * h + s4));
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    const int k = 32 * h + 4 * s4 + ss;
                    const float2 b2 = *reinterpret_cast<const float2*>(Xs + k * XROW32 + (64 * hh + 2 * c) * 4);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b2.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ss], b2.y, acc[1], 0, 0, 0);
                }
            }
        }
    }
    // a wave whose 32 rows x 64 pixels lie inside the tensor reads the addend and writes its rows as straight-line code
    // (as pw_fanin_pipe_kernel: under the row test every addend load is its own round trip, 16 in a row)
    if (ALIGNED && 32 * cb + 32 <= mloc && p0 + PW_TILE <= HW && !(addend && so.partial)) {
        const int px = p0 + 64 * hh + 2 * c;
        T* yw = yn + (long long)(32 * cb) * HW + px;
        if constexpr (Elem<T>::is16) {
            if (addend) {
                const T* aw = addend + ((long long)n * wv.M + m_base + 32 * cb) * HW + px;
                uint32_t ar[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    ar[reg] = *reinterpret_cast<const uint32_t*>(aw + (long long)acc_row(reg, h) * HW);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    T lo, hi;
                    lo.v = (uint16_t)(ar[reg] & 0xffffu);
                    hi.v = (uint16_t)(ar[reg] >> 16);
                    acc[0][reg] += to_float(lo);
                    acc[1][reg] += to_float(hi);
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                *reinterpret_cast<uint32_t*>(yw + (long long)acc_row(reg, h) * HW) = pack2<T>(acc[0][reg], acc[1][reg]);
        } else {
            if (addend) {
                const T* aw = addend + ((long long)n * wv.M + m_base + 32 * cb) * HW + px;
                float2 ar[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    ar[reg] = *reinterpret_cast<const float2*>(aw + (long long)acc_row(reg, h) * HW);
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    acc[0][reg] += ar[reg].x;
                    acc[1][reg] += ar[reg].y;
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                *reinterpret_cast<float2*>(yw + (long long)acc_row(reg, h) * HW) = make_float2(acc[0][reg], acc[1][reg]);
        }
        if (so.partial) stat_epilogue<T, 2>(acc, px, HW, c, h, 32 * cb, mloc, m_base, so, 2 * tile + hh);
        return;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = 32 * cb + acc_row(reg, h);
        if (r < mloc) {
            float v[2] = {acc[0][reg], acc[1][reg]};
            if (addend)
                add_px<T, 2, ALIGNED>(addend + ((long long)n * wv.M + m_base + r) * HW, p0 + 64 * hh + 2 * c, HW, v);
            store_px<T, 2, ALIGNED>(yn + (long long)r * HW, p0 + 64 * hh + 2 * c, HW, v);
        }
    }
    if (so.partial) stat_epilogue<T, 2>(acc, p0 + 64 * hh + 2 * c, HW, c, h, 32 * cb, mloc, m_base, so, 2 * tile + hh);
}

// ---- fan-in, 16-bit aligned, vector-loadable weights: the same tile and chunk loop, software-pipelined.  The X and
// weight chunk of round kc+1 are requested into registers before the MFMAs of round kc, so the LDS fill, the two
// barriers and the MFMAs of a round hide behind the next chunk's HBM round trip instead of adding to it (measured
// timeline of the plain loop: ~10 us of loads + 3.4 us of rounds + W latency, nothing overlapped).
constexpr int FOLD_KMAX = 512;   // input channels a block can fold statistics for (3 LDS tables)
// BX: the X operand is a gradient read through the BN(+ReLU6) backward (BwdXf; needs FAST and K <= FOLD_KMAX)
template <typename T, bool XF, bool FAST = false, bool BX = false>
__global__ void __launch_bounds__(PW_THREADS) pw_fanin_pipe_kernel(const T* __restrict__ x, WView wv, T* __restrict__ y,
                                                                   int HW, int tiles_per_img, int kchunks, InputXf xf,
                                                                   const T* __restrict__ addend, StatOut so,
                                                                   BnFold fold = BnFold{}, BwdXf bx = BwdXf{}) {
    __shared__ __attribute__((aligned(16))) char Ws[64 * WROW16];
    __shared__ __attribute__((aligned(16))) char Xs[64 * XROW16];
    __shared__ float fsc[XF ? FOLD_KMAX : 1], fmu[XF ? FOLD_KMAX : 1], fb[XF ? FOLD_KMAX : 1];
    __shared__ float bxt[BX ? 5 * FOLD_KMAX : 1];   // per input channel: mean | scale | beta' | ka | kbi
    const int tid = threadIdx.x;
    if constexpr (BX) {
        for (int ch = tid; ch < wv.K; ch += PW_THREADS) {
            const float mu = bx.mean[ch], sc = bx.scale[ch];
            bxt[ch] = mu;
            bxt[FOLD_KMAX + ch] = sc;
            bxt[2 * FOLD_KMAX + ch] = fmaf(mu, sc, bx.shift[ch]);
            if (bx.fold_partial) {   // workgroup-uniform: fold the reduction pass's partial slabs here; block (0, 0) publishes
                float ka, kbi;
                bwdxf_fold(bx, ch, sc, blockIdx.x == 0 && blockIdx.y == 0, ka, kbi);
                bxt[3 * FOLD_KMAX + ch] = ka;
                bxt[4 * FOLD_KMAX + ch] = kbi;
            } else {
                bxt[3 * FOLD_KMAX + ch] = bx.ka[ch];
                bxt[4 * FOLD_KMAX + ch] = bx.kbi[ch];
            }
        }
        __syncthreads();
    }
    const bool folded = XF && fold.cp != nullptr;
    if constexpr (XF) {
        if (folded) {   // the input BN's finalize, per block: channel ch from its P partials, fixed order, fp64
            const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
            for (int ch = tid; ch < wv.K; ch += PW_THREADS) {
                double s = 0.0, ss = 0.0;
                // 8 partials per request round (clamped index, so the 8 loads are in flight together), summed in order:
                // the same result as one load per iteration, which is P dependent L2 round trips per channel
                for (int q0 = 0; q0 < fold.P; q0 += 8) {
                    float2 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = q0 + j < fold.P ? q0 + j : fold.P - 1;
                        v[j] = fold.cp[(long long)ch * fold.P + q];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const bool live = q0 + j < fold.P;
                        s += live ? (double)v[j].x : 0.0;
                        ss += live ? (double)v[j].y : 0.0;
                    }
                }
                const double mean = s / fold.M;
                double var = ss / fold.M - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const double invstd = 1.0 / sqrt(var + fold.eps);
                const double g = fold.gamma ? (double)fold.gamma[ch] : 1.0, b = fold.beta ? (double)fold.beta[ch] : 0.0;
                fsc[ch] = (float)(g * invstd);
                fmu[ch] = (float)mean;
                fb[ch] = (float)b;
                if (writer) {
                    fold.mean[ch] = (float)mean;
                    fold.invstd[ch] = (float)invstd;
                    fold.scale[ch] = (float)(g * invstd);
                    fold.shift[ch] = (float)(b - mean * g * invstd);
                    if (fold.running_mean) {
                        const double unb = fold.M > 1.0 ? var * fold.M / (fold.M - 1.0) : var;
                        fold.running_mean[ch] =
                            (float)((1.0 - fold.momentum) * (double)fold.running_mean[ch] + fold.momentum * mean);
                        fold.running_var[ch] =
                            (float)((1.0 - fold.momentum) * (double)fold.running_var[ch] + fold.momentum * unb);
                    }
                }
            }
            __syncthreads();
        }
    }
    const int m_base = blockIdx.y * 64;
    const int mloc = min(64, wv.M - m_base);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int cb = wave & 1, hh = wave >> 1;
    const int tile = blockIdx.x;
    const int n = tile / tiles_per_img;
    const int p0 = (tile - n * tiles_per_img) * PW_TILE;
    const T* xn = x + (long long)n * wv.K * HW;
    const T* yn2 = BX ? reinterpret_cast<const T*>(bx.y) + (long long)n * wv.K * HW : nullptr;
    T* yn = y + ((long long)n * wv.M + m_base) * HW;
    const bool rowmajor = wv.sk == 1;

    uint4 xr[4];
    uint4 yr[BX ? 4 : 1];
    float4 wr[4];
    uint32_t okbits = 0;   // FAST: bit it = weight chunk it is inside the slice, bit 4+it = X vector it is
    // FAST: this lane's element offsets inside a 64-channel chunk (X / y: relative to the chunk's first channel row;
    // weights: relative to the slice's element (k0, 0) resp. (0, k0)) and the chunk-independent halves of the bounds tests
    uint32_t xoff[FAST ? 4 : 1], woff[FAST ? 4 : 1], xpx_ok = 0, wrow_ok = 0;
    if constexpr (FAST) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            const int px = p0 + 8 * (q & 15);
            xoff[it] = (uint32_t)(q >> 4) * (uint32_t)HW + (uint32_t)px;
            xpx_ok |= (px < HW ? 1u : 0u) << it;
            if (rowmajor) {
                const int r = q >> 4;
                woff[it] = (uint32_t)(m_base + r) * (uint32_t)wv.sm + 4u * (q & 15);
                wrow_ok |= (r < mloc ? 1u : 0u) << it;
            } else {
                const int r = 4 * (q & 15);
                woff[it] = (uint32_t)(q >> 4) * (uint32_t)wv.sk + (uint32_t)(m_base + r);
                wrow_ok |= (m_base + r < wv.M ? 1u : 0u) << it;
            }
        }
    }
    auto load_chunk = [&](int kc) {
        const int k0 = 64 * kc;
        if constexpr (FAST) {
            // K % 4 == 0 and M % 4 == 0 (launch conditions): every 16-byte chunk is wholly inside or outside the slice.
            // Branch-free: an outside chunk reads the first chunk instead and is zeroed when it is written to LDS -- a
            // conditional load ends its basic block with s_waitcnt vmcnt(0), which made the 4 weight requests of a
            // round 4 serial L2 round trips
            // Addresses = uniform chunk base + a 32-bit element offset that each lane computes ONCE (xoff / woff below;
            // launch condition: the image and the weight slice span < 2^31 elements): with (long long)k * HW + px formed
            // per request the kernel issued ~70 quarter-rate 32-bit multiplies and as many 64-bit adds per round.
            okbits = 0;
            const float* wk = wv.w + (long long)k0 * (rowmajor ? 1 : wv.sk);              // uniform
            const uint32_t wfirst = (uint32_t)m_base * (rowmajor ? (uint32_t)wv.sm : 1u);   // a chunk that is always inside
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = tid + it * PW_THREADS;
                const bool ok = rowmajor ? ((wrow_ok >> it) & 1u) && k0 + 4 * (q & 15) < wv.K
                                         : ((wrow_ok >> it) & 1u) && k0 + (q >> 4) < wv.K;
                wr[it] = *reinterpret_cast<const float4*>(wk + (ok ? woff[it] : wfirst));
                okbits |= (ok ? 1u : 0u) << it;
            }
            const T* xk = xn + (long long)k0 * HW;                                          // uniform
            const T* yk = BX ? yn2 + (long long)k0 * HW : nullptr;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = tid + it * PW_THREADS;
                const bool ok = ((xpx_ok >> it) & 1u) && k0 + (q >> 4) < wv.K;
                const uint32_t off = ok ? xoff[it] : (uint32_t)p0;
                xr[it] = *reinterpret_cast<const uint4*>(xk + off);
                if constexpr (BX) yr[it] = *reinterpret_cast<const uint4*>(yk + off);
                okbits |= (ok ? 16u : 0u) << it;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {   // weights first: vector-memory returns are in order
            const int q = tid + it * PW_THREADS;
            wr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowmajor) {
                const int r = q >> 4, kk = k0 + 4 * (q & 15);
                if (r < mloc) {
                    const float* src = wv.w + (long long)(m_base + r) * wv.sm + kk;
                    if (kk + 3 < wv.K) wr[it] = *reinterpret_cast<const float4*>(src);
                    else {
                        if (kk < wv.K) wr[it].x = src[0];
                        if (kk + 1 < wv.K) wr[it].y = src[1];
                        if (kk + 2 < wv.K) wr[it].z = src[2];
                    }
                }
            } else {
                const int k = q >> 4, r = 4 * (q & 15);
                const int m = m_base + r, kk = k0 + k;
                if (kk < wv.K) {
                    const float* src = wv.w + (long long)kk * wv.sk + m;
                    if (m + 3 < wv.M) wr[it] = *reinterpret_cast<const float4*>(src);
                    else {
                        if (m < wv.M) wr[it].x = src[0];
                        if (m + 1 < wv.M) wr[it].y = src[1];
                        if (m + 2 < wv.M) wr[it].z = src[2];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            const int k = k0 + (q >> 4), px = p0 + 8 * (q & 15);
            xr[it] = make_uint4(0, 0, 0, 0);
            if (k < wv.K && px < HW) xr[it] = *reinterpret_cast<const uint4*>(xn + (long long)k * HW + px);
        }
    };
    auto store_chunk = [&](int kc) {
        if constexpr (FAST) {
            if (okbits != 0xffu) {   // a lane with every chunk inside (all lanes of an interior tile) skips 32 selects per round
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (!((okbits >> it) & 1u)) wr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (!((okbits >> (4 + it)) & 1u)) xr[it] = make_uint4(0, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            if (rowmajor) {
                const int r = q >> 4, k = 4 * (q & 15);
                *reinterpret_cast<uint2*>(Ws + wtile_off<T>(r, k)) =
                    make_uint2(pack2<T>(wr[it].x, wr[it].y), pack2<T>(wr[it].z, wr[it].w));
            } else {
                const int k = q >> 4, r = 4 * (q & 15);
                lds_store_w<T>(Ws, wtile_off<T>(r, k), wr[it].x);
                lds_store_w<T>(Ws, wtile_off<T>(r + 1, k), wr[it].y);
                lds_store_w<T>(Ws, wtile_off<T>(r + 2, k), wr[it].z);
                lds_store_w<T>(Ws, wtile_off<T>(r + 3, k), wr[it].w);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int q = tid + it * PW_THREADS;
            const int k = q >> 4, m = q & 15;
            uint4 v = xr[it];
            if constexpr (XF) {
                if (64 * kc + k < wv.K && p0 + 8 * m < HW) {
                    const int ch = 64 * kc + k;
                    v = folded ? xf_apply8_core<T>(v, fsc[ch], fmu[ch], fb[ch]) : xf_apply8<T>(v, xf, ch);
                }
            }
            if constexpr (BX) {   // chunks outside the slice were zeroed above and stay zero (ka of a dead chunk is not applied)
                if ((okbits >> (4 + it)) & 1u) {
                    const int ch = 64 * kc + k;
                    v = bx_apply8<T>(v, yr[it], bxt[ch], bxt[FOLD_KMAX + ch], bxt[2 * FOLD_KMAX + ch], bxt[3 * FOLD_KMAX + ch],
                                     bxt[4 * FOLD_KMAX + ch]);
                    // the tile's dy, once (the blocks of the first output-row slab), for the weight-gradient kernel
                    if (bx.dy_out && blockIdx.y == 0)   // same element offset as the request: uniform chunk base + xoff
                        *reinterpret_cast<uint4*>(reinterpret_cast<T*>(bx.dy_out) + ((long long)n * wv.K + 64 * kc) * HW + xoff[it]) = v;
                }
            }
            // pixel 8m+i -> position 64*(m>>3) + 32*(i&1) + 4*(m&7) + (i>>1)   (stage_x_tile, PX = 2)
            char* rowp = Xs + k * XROW16;
            const int base = 64 * (m >> 3) + 4 * (m & 7);
            *reinterpret_cast<uint2*>(rowp + base * 2) =
                make_uint2((v.x & 0xffffu) | (v.y << 16), (v.z & 0xffffu) | (v.w << 16));
            *reinterpret_cast<uint2*>(rowp + (base + 32) * 2) =
                make_uint2((v.x >> 16) | (v.y & 0xffff0000u), (v.z >> 16) | (v.w & 0xffff0000u));
        }
    };

    f32x16 acc[2];
    acc[0] = zero16();
    acc[1] = zero16();
    const int row = 32 * cb + c;
    auto mma_round = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const s16x8 af = *reinterpret_cast<const s16x8*>(Ws + wtile_chunk_off<T>(row, 2 * s + h));
            const s16x8 b0 = read_b_frag16(Xs, 64 * hh, s, lane);
            const s16x8 b1 = read_b_frag16(Xs, 64 * hh + 32, s, lane);
            acc[0] = Mma16<T>::run(af, b0, acc[0]);
            acc[1] = Mma16<T>::run(af, b1, acc[1]);
        }
    };
    load_chunk(0);
    for (int kc = 0; kc < kchunks; ++kc) {
        if (kc) __syncthreads();   // the previous round's readers are done with Ws / Xs
        store_chunk(kc);
        __syncthreads();
        if (kc + 1 < kchunks) load_chunk(kc + 1);   // in flight during this round's MFMAs and the next barrier
        mma_round();
    }
    // a wave whose 32 rows x 64 pixels lie inside the tensor writes them as straight-line code: a store (or the addend's
    // load) under a lane- or row-dependent branch ends its basic block with s_waitcnt vmcnt(0), i.e. 16 serial round trips
    if (32 * cb + 32 <= mloc && p0 + PW_TILE <= HW && !(addend && so.partial)) {
        const int px = p0 + 64 * hh + 2 * c;
        T* yw = yn + (long long)(32 * cb) * HW + px;
        if (addend) {
            const T* aw = addend + ((long long)n * wv.M + m_base + 32 * cb) * HW + px;
            uint32_t ar[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                ar[reg] = *reinterpret_cast<const uint32_t*>(aw + (long long)acc_row(reg, h) * HW);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                T lo, hi;
                lo.v = (uint16_t)(ar[reg] & 0xffffu);
                hi.v = (uint16_t)(ar[reg] >> 16);
                acc[0][reg] += to_float(lo);
                acc[1][reg] += to_float(hi);
            }
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            *reinterpret_cast<uint32_t*>(yw + (long long)acc_row(reg, h) * HW) = pack2<T>(acc[0][reg], acc[1][reg]);
        // (statistics are of the conv output proper: with an addend as well the generic path below runs)
        if (so.partial) stat_epilogue<T, 2>(acc, px, HW, c, h, 32 * cb, mloc, m_base, so, 2 * tile + hh);
        return;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int r = 32 * cb + acc_row(reg, h);
        if (r < mloc) {
            float v[2] = {acc[0][reg], acc[1][reg]};
            if (addend)
                add_px<T, 2, true>(addend + ((long long)n * wv.M + m_base + r) * HW, p0 + 64 * hh + 2 * c, HW, v);
            store_px<T, 2, true>(yn + (long long)r * HW, p0 + 64 * hh + 2 * c, HW, v);
        }
    }
    if (so.partial) stat_epilogue<T, 2>(acc, p0 + 64 * hh + 2 * c, HW, c, h, 32 * cb, mloc, m_base, so, 2 * tile + hh);
}

// -------------------------------------------------------------------------------------- wgrad
// out(r, s) = sum_{n,p} R[n][r][p] * S[n][s][p];  R: [N][MR][HW], S: [N][NS][HW].
// block: 64 R-rows x 64 S-rows, one 32x32 MFMA tile per wave; grid.z splits the (n, pixel-chunk) range;
// partials [z][MR][NS] fp32 are summed by pw_wgrad_reduce_kernel in a fixed order (deterministic).
template <typename T, bool ALIGNED>
__device__ __forceinline__ void stage_rows64(char* Lt, const T* __restrict__ base, int rows_total, int r0, int HW,
                                             int p0) {
    // [64 rows][64 px] -> operand tile (row-major, px = k)
    const int tid = threadIdx.x;
    if constexpr (Elem<T>::is16) {
        if (ALIGNED) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int q = tid + it * PW_THREADS;  // 64 rows x 8 chunks
                const int r = q >> 3, ch = q & 7;
                const int px = p0 + 8 * ch;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r0 + r < rows_total && px < HW)
                    v = *reinterpret_cast<const uint4*>(base + (long long)(r0 + r) * HW + px);
                *reinterpret_cast<uint4*>(Lt + wtile_chunk_off<T>(r, ch)) = v;
            }
        } else {
            for (int e = tid; e < 64 * 64; e += PW_THREADS) {
                const int r = e >> 6, k = e & 63;
                uint16_t v = 0;
                if (r0 + r < rows_total && p0 + k < HW)
                    v = reinterpret_cast<const uint16_t*>(base)[(long long)(r0 + r) * HW + p0 + k];
                *reinterpret_cast<uint16_t*>(Lt + wtile_off<T>(r, k)) = v;
            }
        }
    } else {
        if (ALIGNED) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int q = tid + it * PW_THREADS;  // 64 rows x 16 chunks
                const int r = q >> 4, ch = q & 15;
                const int px = p0 + 4 * ch;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r0 + r < rows_total && px < HW)
                    v = *reinterpret_cast<const uint4*>(base + (long long)(r0 + r) * HW + px);
                *reinterpret_cast<uint4*>(Lt + wtile_chunk_off<T>(r, ch)) = v;
            }
        } else {
            for (int e = tid; e < 64 * 64; e += PW_THREADS) {
                const int r = e >> 6, k = e & 63;
                float v = 0.f;
                if (r0 + r < rows_total && p0 + k < HW)
                    v = reinterpret_cast<const float*>(base)[(long long)(r0 + r) * HW + p0 + k];
                *reinterpret_cast<float*>(Lt + wtile_off<T>(r, k)) = v;
            }
        }
    }
}

// pixels staged per barrier: 4 (16-bit) / 2 (fp32) swizzled [64][64] sub-tiles per operand = 64 KiB of LDS,
// i.e. 64 KiB of loads in flight per block -- the reduction has almost no arithmetic per byte (one MFMA per
// 8 KiB), so it only runs at HBM speed if every barrier interval moves a lot of data.
template <typename T> struct WgStage { static constexpr int SUB = Elem<T>::is16 ? 4 : 2; };

// The aligned staging in two halves: rows64_request issues the [64 rows][64 px] sub-tile's 16-byte requests branch-free
// (a chunk outside the tensor reads the image's first chunk instead and is zeroed on deposit); rows64_deposit writes
// them to the operand tile.  With the load under its bounds test -- stage_rows64 -- every request ended its basic block
// with s_waitcnt vmcnt(0): the 16 requests of a stage were 16 serial round trips (fp32 expand gradient: 120 us for
// 118 MB).
template <typename T> struct Rows64 {
    static constexpr int NIT = Elem<T>::is16 ? 2 : 4;     // 16-byte chunks per thread
    static constexpr int CPR = Elem<T>::is16 ? 8 : 16;    // chunks per row
    static constexpr int EPC = Elem<T>::is16 ? 8 : 4;     // elements per chunk
    uint4 v[NIT];
    uint32_t ok;
};
template <typename T>
__device__ __forceinline__ void rows64_request(Rows64<T>& g, const T* __restrict__ base, int rows_total, int r0, int HW, int p0) {
    const int tid = threadIdx.x;
    g.ok = 0;
#pragma unroll
    for (int it = 0; it < Rows64<T>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        const int r = q / Rows64<T>::CPR, ch = q % Rows64<T>::CPR;
        const int px = p0 + Rows64<T>::EPC * ch;
        const bool ok = r0 + r < rows_total && px < HW;
        g.v[it] = *reinterpret_cast<const uint4*>(base + (ok ? (long long)(r0 + r) * HW + px : 0));
        g.ok |= (ok ? 1u : 0u) << it;
    }
}
template <typename T>
__device__ __forceinline__ void rows64_deposit(char* Lt, const Rows64<T>& g) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < Rows64<T>::NIT; ++it) {
        const int q = tid + it * PW_THREADS;
        const int r = q / Rows64<T>::CPR, ch = q % Rows64<T>::CPR;
        const uint32_t m = (g.ok >> it) & 1u ? 0xffffffffu : 0u;
        *reinterpret_cast<uint4*>(Lt + wtile_chunk_off<T>(r, ch)) = make_uint4(g.v[it].x & m, g.v[it].y & m, g.v[it].z & m, g.v[it].w & m);
    }
}

template <typename T, bool ALIGNED>
__global__ void __launch_bounds__(PW_THREADS) pw_wgrad_kernel(const T* __restrict__ R, const T* __restrict__ S,
                                                              float* __restrict__ part, int MR, int NS, int HW,
                                                              int stages_per_img, int total_stages,
                                                              int stages_per_split) {
    constexpr int SUB = WgStage<T>::SUB;
    constexpr int TB = 64 * Elem<T>::wrow;   // bytes of one [64][64] sub-tile
    __shared__ __attribute__((aligned(16))) char Rt[SUB * TB];
    __shared__ __attribute__((aligned(16))) char St[SUB * TB];
    const int r0 = blockIdx.x * 64, s0 = blockIdx.y * 64, z = blockIdx.z;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 31, h = lane >> 5;
    const int rb = wave & 1, sb = wave >> 1;
    f32x16 acc = zero16();
    const int q_lo = z * stages_per_split;
    const int q_hi = min(total_stages, q_lo + stages_per_split);
    auto multiply = [&]() {
#pragma unroll
        for (int sub = 0; sub < SUB; ++sub) {
            const char* Rs = Rt + sub * TB;
            const char* Ss = St + sub * TB;
            if constexpr (Elem<T>::is16) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const s16x8 af = *reinterpret_cast<const s16x8*>(Rs + wtile_chunk_off<T>(32 * rb + c, 2 * s + h));
                    const s16x8 bf = *reinterpret_cast<const s16x8*>(Ss + wtile_chunk_off<T>(32 * sb + c, 2 * s + h));
                    acc = Mma16<T>::run(af, bf, acc);
                }
            } else {
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    const float4 a4 = *reinterpret_cast<const float4*>(Rs + wtile_chunk_off<T>(32 * rb + c, 8 * h + s4));
                    const float4 b4 = *reinterpret_cast<const float4*>(Ss + wtile_chunk_off<T>(32 * sb + c, 8 * h + s4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
                }
            }
        }
    };
    if constexpr (ALIGNED) {
        // stage q + 1 is requested into registers (64 of them) before stage q is multiplied: its round trip runs
        // under the MFMAs and the barrier instead of after them
        Rows64<T> rg[SUB], sg[SUB];
        auto request = [&](int q) {
            const int n = q / stages_per_img;
            const int p0 = (q - n * stages_per_img) * (64 * SUB);
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                rows64_request<T>(rg[sub], R + (long long)n * MR * HW, MR, r0, HW, p0 + 64 * sub);
                rows64_request<T>(sg[sub], S + (long long)n * NS * HW, NS, s0, HW, p0 + 64 * sub);
            }
        };
        if (q_lo < q_hi) request(q_lo);
        for (int q = q_lo; q < q_hi; ++q) {
            if (q > q_lo) __syncthreads();
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                rows64_deposit<T>(Rt + sub * TB, rg[sub]);
                rows64_deposit<T>(St + sub * TB, sg[sub]);
            }
            __syncthreads();
            if (q + 1 < q_hi) request(q + 1);
            multiply();
        }
    } else {
        for (int q = q_lo; q < q_hi; ++q) {
            const int n = q / stages_per_img;
            const int p0 = (q - n * stages_per_img) * (64 * SUB);
            if (q > q_lo) __syncthreads();
#pragma unroll
            for (int sub = 0; sub < SUB; ++sub) {
                stage_rows64<T, ALIGNED>(Rt + sub * TB, R + (long long)n * MR * HW, MR, r0, HW, p0 + 64 * sub);
                stage_rows64<T, ALIGNED>(St + sub * TB, S + (long long)n * NS * HW, NS, s0, HW, p0 + 64 * sub);
            }
            __syncthreads();
            multiply();
        }
    }
    // D[row = R row][col = S row]: lane holds column c, 16 rows
    float* pz = part + (long long)z * MR * NS;
    const int sc = s0 + 32 * sb + c;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int rr = r0 + 32 * rb + acc_row(reg, h);
        if (rr < MR && sc < NS) pz[(long long)rr * NS + sc] = acc[reg];
    }
}

// dw[r*sr + s*ss] = sum_z part[z][r][s]: 64 outputs x 16 z-lanes per block (each z-lane sums every 16th slab with 4
// independent loads in flight), z-lane partial sums combined through LDS in a fixed order (deterministic)
constexpr int WR_ZL = 16;
__global__ void __launch_bounds__(64 * WR_ZL) pw_wgrad_reduce_kernel(const float* __restrict__ part,
                                                                     float* __restrict__ dw, int MR, int NS, int nsplit,
                                                                     long long sr, long long ss) {
    __shared__ float red[WR_ZL][64];
    const long long tot = (long long)MR * NS;
    const long long idx = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int zl = threadIdx.x >> 6;
    float a = 0.f;
    if (idx < tot) {
        int z = zl;
        for (; z + 3 * WR_ZL < nsplit; z += 4 * WR_ZL) {
            const float v0 = part[(long long)z * tot + idx], v1 = part[(long long)(z + WR_ZL) * tot + idx];
            const float v2 = part[(long long)(z + 2 * WR_ZL) * tot + idx], v3 = part[(long long)(z + 3 * WR_ZL) * tot + idx];
            a += (v0 + v1) + (v2 + v3);
        }
        for (; z < nsplit; z += WR_ZL) a += part[(long long)z * tot + idx];
    }
    red[zl][threadIdx.x & 63] = a;
    __syncthreads();
    if (zl == 0 && idx < tot) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WR_ZL; ++i) t += red[i][threadIdx.x];
        const long long r = idx / NS, sx = idx - r * NS;
        dw[r * sr + sx * ss] = t;
    }
}

// ---- 16-bit aligned fast path: no LDS, no barrier (measured 24.6 us + 7.2 us reduce vs 33.1 + 5.0 for the staged
// kernel on 16x384x64x64; per-lane 16-byte loads of 32 different lines run at ~1 lane/clk in the address path, so
// the kernel wants as many blocks as CUs).  With k = pixels BOTH operands of out(r, s) = sum_px R[r][px]*S[s][px]
// have their 8 k-values contiguous in NCHW, so a fragment is a plain 16-byte global load.  The k-slot <-> pixel map is
// free as long as A and B agree: lane (row, h) owns the 32 pixels [64q + 32h, +32) of quad q -- 64 contiguous bytes,
// consumed by the quad's 4 k-steps -- so the two lanes of a row cover one full 128-byte line.
// block = 8 waves = 4 row groups (3 row blocks each: 384 rows) x 2 column blocks (64 cols); grid.x = split-K.
constexpr int WD_RB = 3;
constexpr int WD_ROWS = 4 * WD_RB * 32, WD_COLS = 64;

struct WdPlan {
    int quads_per_img, total_quads, nsplit, MR, NS;
};

// XF: 0 none, 1 the R operand, 2 the S operand is read through the fused BN + ReLU6 (InputXf); 3: the R operand is the
// gradient of a BN(+ReLU6) OUTPUT and the product wants the gradient of its INPUT: dy is formed from (R = da, bx.y) with
// the finished per-channel coefficients bx.ka / bx.kbi as the fragments are read (BwdXf; the expand weight gradient --
// the chain's expand input gradient then stores no dy1 for this kernel)
template <typename T, int XF = 0>
__global__ void __launch_bounds__(512) pw_wgrad_direct_kernel(const T* __restrict__ R, const T* __restrict__ S,
                                                              float* __restrict__ part, int MR, int NS, int HW,
                                                              WdPlan wp, InputXf xf = InputXf{}, BwdXf bx = BwdXf{}) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave & 3, wc = wave >> 2;
    const int c = lane & 31, h = lane >> 5;
    const int split = blockIdx.x;
    const int row0 = blockIdx.y * WD_ROWS + wr * (WD_RB * 32);   // first row of this wave
    const int col = blockIdx.z * WD_COLS + wc * 32 + c;          // this lane's S channel
    if (row0 >= MR) return;   // no rows (and no slab rows) for this wave; the kernel has no barrier
    f32x16 acc[WD_RB];
#pragma unroll
    for (int rb = 0; rb < WD_RB; ++rb) acc[rb] = zero16();
    const int q0 = (int)((long long)split * wp.total_quads / wp.nsplit);
    const int q1 = (int)((long long)(split + 1) * wp.total_quads / wp.nsplit);
    const bool cs = col < NS;
    const int colc = cs ? col : NS - 1;                          // clamped: every request goes to a valid address
    bool rs[WD_RB];
    int rowc[WD_RB];
#pragma unroll
    for (int rb = 0; rb < WD_RB; ++rb) {
        rs[rb] = row0 + 32 * rb + c < MR;
        rowc[rb] = rs[rb] ? row0 + 32 * rb + c : MR - 1;
    }
    // fused transforms: this lane's rows / column are fixed, so are its channel constants
    float xsc[WD_RB], xmu[WD_RB], xb[WD_RB];
    float csc = 1.f, cmu = 0.f, cbb = 0.f;
    if constexpr (XF == 1) {
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb) {
            xsc[rb] = xf.scale[rowc[rb]];
            xmu[rb] = xf.mean[rowc[rb]];
            xb[rb] = fmaf(xmu[rb], xsc[rb], xf.shift[rowc[rb]]);
        }
    }
    if constexpr (XF == 2) {
        csc = xf.scale[colc];
        cmu = xf.mean[colc];
        cbb = fmaf(cmu, csc, xf.shift[colc]);
    }
    float bka[WD_RB], bkbi[WD_RB];
    if constexpr (XF == 3) {
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb) {
            xsc[rb] = bx.scale[rowc[rb]];
            xmu[rb] = bx.mean[rowc[rb]];
            xb[rb] = fmaf(xmu[rb], xsc[rb], bx.shift[rowc[rb]]);
            bka[rb] = bx.ka[rowc[rb]];
            bkbi[rb] = bx.kbi[rowc[rb]];
        }
    }
    long long rowoff[WD_RB];   // element offset of this lane's rows / column inside an image
#pragma unroll
    for (int rb = 0; rb < WD_RB; ++rb) rowoff[rb] = (long long)rowc[rb] * HW;
    const long long coloff = (long long)colc * HW;
    for (int q = q0; q < q1; ++q) {
        const int n = q / wp.quads_per_img;
        const int px = (q - n * wp.quads_per_img) * 64 + 32 * h;   // this lane's first pixel
        // Straight-line requests from clamped addresses, dead lanes / chunks zeroed afterwards: a load under a lane-
        // dependent branch ends its basic block with s_waitcnt vmcnt(0), which had made the 16 requests of a quad as many
        // serial round trips
        uint4 a[WD_RB][4], b[4];
        bool okp[4];
        int pxc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            okp[j] = px + 8 * j < HW;
            pxc[j] = okp[j] ? px + 8 * j : 0;
        }
        // uniform image base + the lane's row offset (formed once, before the loop) + pixel: the 64-bit
        // (n * MR + row) * HW per request was ~16 quarter-rate multiplies per quad beside its 12 MFMAs
        const T* sp = S + (long long)n * NS * HW + coloff;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const uint4*>(sp + pxc[j]);
        const T* rn = R + (long long)n * MR * HW;
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[rb][j] = *reinterpret_cast<const uint4*>(rn + rowoff[rb] + pxc[j]);
            }
        }
        uint4 ya[XF == 3 ? WD_RB : 1][4];
        if constexpr (XF == 3) {
            const T* yn = reinterpret_cast<const T*>(bx.y) + (long long)n * MR * HW;
#pragma unroll
            for (int rb = 0; rb < WD_RB; ++rb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ya[rb][j] = *reinterpret_cast<const uint4*>(yn + rowoff[rb] + pxc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (XF == 2) b[j] = xf_apply8_core<T>(b[j], csc, cmu, cbb);
            if (!(cs && okp[j])) b[j] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (XF == 1) a[rb][j] = xf_apply8_core<T>(a[rb][j], xsc[rb], xmu[rb], xb[rb]);
                if constexpr (XF == 3) a[rb][j] = bx_apply8<T>(a[rb][j], ya[rb][j], xmu[rb], xsc[rb], xb[rb], bka[rb], bkbi[rb]);
                if (!(rs[rb] && okp[j])) a[rb][j] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < WD_RB; ++rb)
                acc[rb] = Mma16<T>::run(__builtin_bit_cast(s16x8, a[rb][j]), __builtin_bit_cast(s16x8, b[j]), acc[rb]);
    }
    float* dst = part + (long long)split * MR * NS;
    if (cs) {
        // rows row0 + 4h + {0..3, 8..11, 16..19, ...}: a running pointer (+NS, +NS, +NS, +5 NS) instead of a 64-bit r * NS
        // per store (96 quarter-rate multiplies per wave for a loop of 4 quads)
        float* p = dst + (long long)(row0 + 4 * h) * NS + col;
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + 32 * rb + acc_row(reg, h);
                if (r < MR) *p = acc[rb][reg];
                p += (reg & 3) == 3 ? 5 * (long long)NS : (long long)NS;
            }
    }
}

static WdPlan wd_plan(int64_t N, int64_t Cin, int64_t Cout, int64_t HW) {
    WdPlan p;
    p.MR = (int)(Cout >= Cin ? Cout : Cin);
    p.NS = (int)(Cout >= Cin ? Cin : Cout);
    p.quads_per_img = (int)cdiv(HW, 64);
    p.total_quads = (int)(N * p.quads_per_img);
    const int64_t tiles = cdiv(p.MR, WD_ROWS) * cdiv(p.NS, WD_COLS);
    static const int blocks = [] { const char* e = getenv("OFASR_PW_WGRAD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    int64_t want = blocks / (tiles > 0 ? tiles : 1);   // one 8-wave block per CU ...
    if (want > p.total_quads / 4) want = p.total_quads / 4;   // ... but at least 4 quads per slab written
    if (want < 1) want = 1;
    p.nsplit = (int)want;
    return p;
}


// ---- fp32 direct weight gradient (round 3): the scheme of pw_wgrad_direct_kernel on v_mfma_f32_32x32x2_f32.  A lane
// (row, h) owns 16 consecutive pixels [32 u + 16 h, +16) of a 32-pixel unit -- four 16-byte requests per operand row,
// straight from global memory, no LDS and no barrier -- and the unit's 16 matrix steps take element j of both lanes' runs
// (k = h <-> pixel 32 u + 16 h + j: any bijection works as long as both operands use it).  The requests of unit u + 1
// are issued before the 48 matrix instructions of unit u.  The staged kernel it replaces (LDS tiles, two barriers per
// 128 pixels) ran at 123 us for the 384 x 64 gradient of a [16, ., 64, 64] block; the matrix floor is 20 us.
struct WdPlan32 {
    int units_per_img, total_units, nsplit, MR, NS;
};

__global__ void __launch_bounds__(512) pw_wgrad_direct_f32_kernel(const float* __restrict__ R, const float* __restrict__ S,
                                                                  float* __restrict__ part, int MR, int NS, int HW,
                                                                  WdPlan32 wp) {
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave & 3, wc = wave >> 2;
    const int c = lane & 31, h = lane >> 5;
    const int split = blockIdx.x;
    const int row0 = blockIdx.y * WD_ROWS + wr * (WD_RB * 32);
    const int col = blockIdx.z * WD_COLS + wc * 32 + c;
    if (row0 >= MR) return;   // (no barrier in this kernel)
    f32x16 acc[WD_RB];
#pragma unroll
    for (int rb = 0; rb < WD_RB; ++rb) acc[rb] = zero16();
    const int u0 = (int)((long long)split * wp.total_units / wp.nsplit);
    const int u1 = (int)((long long)(split + 1) * wp.total_units / wp.nsplit);
    const bool cs = col < NS;
    const int colc = cs ? col : NS - 1;
    bool rs[WD_RB];
    long long rowoff[WD_RB];
#pragma unroll
    for (int rb = 0; rb < WD_RB; ++rb) {
        rs[rb] = row0 + 32 * rb + c < MR;
        rowoff[rb] = (long long)(rs[rb] ? row0 + 32 * rb + c : MR - 1) * HW;
    }
    float4 a[2][WD_RB][4], b[2][4];
    auto request = [&](int set, int u) {
        const int uc = u < u1 ? u : u1 - 1;                   // beyond the range: re-request the last unit (contributes zero)
        const int n = uc / wp.units_per_img;
        const int px = (uc - n * wp.units_per_img) * 32 + 16 * h;
        int pxc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pxc[j] = px + 4 * j < HW ? px + 4 * j : 0;
        const float* sp = S + ((long long)n * NS + colc) * HW;
        const float* rp = R + (long long)n * MR * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[set][j] = *reinterpret_cast<const float4*>(sp + pxc[j]);
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[set][rb][j] = *reinterpret_cast<const float4*>(rp + rowoff[rb] + pxc[j]);
    };
    auto multiply = [&](int set, int u) {
        const int n = u / wp.units_per_img;
        const int px = (u - n * wp.units_per_img) * 32 + 16 * h;
        const bool live = u < u1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = live && cs && px + 4 * j < HW;
            const float4 bv = ok ? b[set][j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int rb = 0; rb < WD_RB; ++rb) {
                const float4 av = rs[rb] ? a[set][rb][j] : make_float4(0.f, 0.f, 0.f, 0.f);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[rb], 0, 0, 0);
            }
        }
    };
    if (u0 < u1) {
        request(0, u0);
        for (int u = u0; u < u1; u += 2) {     // whole pairs: straight-line body, two named register sets
            request(1, u + 1);
            multiply(0, u);
            request(0, u + 2);
            multiply(1, u + 1);
        }
    }
    float* dst = part + (long long)split * MR * NS;
    if (cs) {
        // rows row0 + 4h + {0..3, 8..11, 16..19, ...}: a running pointer (+NS, +NS, +NS, +5 NS) instead of a 64-bit r * NS
        // per store (96 quarter-rate multiplies per wave for a loop of 4 quads)
        float* p = dst + (long long)(row0 + 4 * h) * NS + col;
#pragma unroll
        for (int rb = 0; rb < WD_RB; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + 32 * rb + acc_row(reg, h);
                if (r < MR) *p = acc[rb][reg];
                p += (reg & 3) == 3 ? 5 * (long long)NS : (long long)NS;
            }
    }
}

static WdPlan32 wd_plan32(int64_t N, int64_t Cin, int64_t Cout, int64_t HW) {
    WdPlan32 p;
    p.MR = (int)(Cout >= Cin ? Cout : Cin);
    p.NS = (int)(Cout >= Cin ? Cin : Cout);
    p.units_per_img = (int)cdiv(HW, 32);
    p.total_units = (int)(N * p.units_per_img);
    const int64_t tiles = cdiv(p.MR, WD_ROWS) * cdiv(p.NS, WD_COLS);
    int64_t want = 256 / (tiles > 0 ? tiles : 1);   // one 8-wave block per CU ...
    if (want > p.total_units / 8) want = p.total_units / 8;   // ... but at least 8 units per slab written
    if (want < 1) want = 1;
    p.nsplit = (int)want;
    return p;
}

struct WgradPlan {
    int stages_per_img, total_stages, nsplit, stages_per_split, MR, NS;
};

static WgradPlan wgrad_plan(int64_t N, int64_t Cin, int64_t Cout, int64_t HW, int sub) {
    WgradPlan p;
    p.MR = (int)(Cout >= Cin ? Cout : Cin);
    p.NS = (int)(Cout >= Cin ? Cin : Cout);
    p.stages_per_img = (int)cdiv(HW, 64 * sub);
    p.total_stages = (int)(N * p.stages_per_img);
    const int64_t tiles = cdiv(p.MR, 64) * cdiv(p.NS, 64);
    int64_t want = 512 / (tiles > 0 ? tiles : 1);   // ~2 blocks per CU
    if (want < 1) want = 1;
    if (want > p.total_stages) want = p.total_stages;
    if (want < 1) want = 1;
    p.stages_per_split = (int)cdiv(p.total_stages, want);
    if (p.stages_per_split < 1) p.stages_per_split = 1;
    p.nsplit = (int)cdiv(p.total_stages, p.stages_per_split);
    if (p.nsplit < 1) p.nsplit = 1;
    return p;
}

static bool aligned_for(const void* a, const void* b, int64_t HW, bool is16) {
    const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    return (bits & 15) == 0 && (HW % (is16 ? 8 : 4)) == 0;
}

template <typename T, bool AL, bool WV, bool XF = false>
static void launch_gemm_v(const void* x, WView wv, void* y, int64_t HW, int tiles_per_img, int total_tiles,
                          hipStream_t st, InputXf xf = InputXf{}, const void* addend = nullptr,
                          StatOut so = StatOut{nullptr, 0}, BnFold fold = BnFold{}) {
    if (wv.K <= 64) {
        if constexpr (Elem<T>::is16 && AL && WV && !XF) {
            // one block per pixel tile walking the row slabs (OFASR_PW_FANOUT_SLABS=0: per-slab grid)
            static const bool slabs = [] { const char* e = getenv("OFASR_PW_FANOUT_SLABS"); return !(e && e[0] == '0'); }();
            const int nslab = wv.M / FO_ROWS;
            if (slabs && addend == nullptr && wv.K == 64 && wv.M % FO_ROWS == 0 && (nslab == 2 || nslab == 3) &&
                HW % PW_TILE == 0) {
                dim3 g1((unsigned)total_tiles);
#define OFASR_FO_SLABS(NS, ST)                                                                                       \
    OFASR_LAUNCH((pw_fanout_slabs_kernel<T, NS, ST>), g1, dim3(PW_THREADS), 0, st, (const T*)x, wv, (T*)y, (int)HW, \
                       tiles_per_img, so, BwdStatOut{})
                if (nslab == 3) { if (so.partial) OFASR_FO_SLABS(3, 1); else OFASR_FO_SLABS(3, 0); }
                else { if (so.partial) OFASR_FO_SLABS(2, 1); else OFASR_FO_SLABS(2, 0); }
#undef OFASR_FO_SLABS
                return;
            }
        }
        dim3 grid((unsigned)total_tiles, (unsigned)cdiv(wv.M, FO_ROWS));
        OFASR_LAUNCH((pw_fanout_kernel<T, AL, WV, XF>), grid, dim3(PW_THREADS), 0, st, (const T*)x, wv, (T*)y,
                           (int)HW, tiles_per_img, xf, (const T*)addend, so);
    } else {
        dim3 grid((unsigned)total_tiles, (unsigned)cdiv(wv.M, 64));
        if constexpr (Elem<T>::is16 && AL && WV) {
            static const bool fast_ok = [] { const char* e = getenv("OFASR_PW_FANIN_FAST"); return !(e && e[0] == '0'); }();
            const bool span32 = (long long)wv.K * HW < (1LL << 31) &&
                                (long long)wv.M * wv.sm + (long long)wv.K * wv.sk < (1LL << 31);   // 32-bit lane offsets
            if (fast_ok && wv.K % 4 == 0 && wv.M % 4 == 0 && span32)
                OFASR_LAUNCH((pw_fanin_pipe_kernel<T, XF, true>), grid, dim3(PW_THREADS), 0, st, (const T*)x, wv,
                                   (T*)y, (int)HW, tiles_per_img, (int)cdiv(wv.K, 64), xf, (const T*)addend, so, fold);
            else
                OFASR_LAUNCH((pw_fanin_pipe_kernel<T, XF, false>), grid, dim3(PW_THREADS), 0, st, (const T*)x, wv,
                                   (T*)y, (int)HW, tiles_per_img, (int)cdiv(wv.K, 64), xf, (const T*)addend, so, fold);
            return;
        }
        OFASR_LAUNCH((pw_fanin_kernel<T, AL, WV, XF>), grid, dim3(PW_THREADS), 0, st, (const T*)x, wv, (T*)y,
                           (int)HW, tiles_per_img, (int)cdiv(wv.K, 64), xf, (const T*)addend, so);
    }
}

template <typename T, bool XF = false>
static int launch_gemm(const char* name, const void* x, WView wv, void* y, int64_t N, int64_t HW, hipStream_t st,
                       InputXf xf = InputXf{}, const void* addend = nullptr, StatOut so = StatOut{nullptr, 0},
                       BnFold fold = BnFold{}) {
    constexpr bool is16 = Elem<T>::is16;
    const bool al = aligned_for(x, y, HW, is16);
    const long long ld = wv.sk == 1 ? wv.sm : wv.sk;
    // vector weight staging: 16-byte aligned rows AND K, M multiples of 4, so that a chunk never straddles the slice
    const bool wvec = (reinterpret_cast<uintptr_t>(wv.w) & 15) == 0 && (ld % 4) == 0 && wv.K % 4 == 0 && wv.M % 4 == 0;
    const int tiles_per_img = (int)cdiv(HW, PW_TILE);
    const int64_t total64 = N * tiles_per_img;
    OFASR_REQUIRE(total64 <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "%s: too many pixel tiles", name);
    const int total_tiles = (int)total64;
    {   // algorithmic bytes / flops of this launch: read K, write M channels of N*HW pixels
        const double px = (double)N * (double)HW, es = (double)sizeof(T);
        prof_note(es * px * (wv.K + wv.M + (addend ? wv.M : 0)) + 4.0 * wv.K * wv.M, 2.0 * px * wv.K * wv.M);
    }
    if (so.partial)
        OFASR_REQUIRE(so.P == (wv.K <= 64 ? total_tiles : 2 * total_tiles), OFASR_ERR_INVALID_ARG,
                      "%s: statistics unit count %d does not match the launch", name, so.P);
    if constexpr (XF) {
        OFASR_REQUIRE(al && is16, OFASR_ERR_UNSUPPORTED, "%s: fused input transform needs aligned 16-bit tensors", name);
        if (fold.cp)
            OFASR_REQUIRE(wvec && wv.K > 64 && wv.K <= FOLD_KMAX, OFASR_ERR_UNSUPPORTED,
                          "%s: folded statistics need the pipelined fan-in kernel (64 < K <= %d)", name, FOLD_KMAX);
        if (wvec) launch_gemm_v<T, true, true, true>(x, wv, y, HW, tiles_per_img, total_tiles, st, xf, nullptr, so, fold);
        else launch_gemm_v<T, true, false, true>(x, wv, y, HW, tiles_per_img, total_tiles, st, xf, nullptr, so);
        return check_launch(name);
    }
    const bool al2 = al && (addend == nullptr || (reinterpret_cast<uintptr_t>(addend) & 15) == 0);
    if (al2 && wvec) launch_gemm_v<T, true, true>(x, wv, y, HW, tiles_per_img, total_tiles, st, InputXf{}, addend, so);
    else if (al2) launch_gemm_v<T, true, false>(x, wv, y, HW, tiles_per_img, total_tiles, st, InputXf{}, addend, so);
    else if (wvec) launch_gemm_v<T, false, true>(x, wv, y, HW, tiles_per_img, total_tiles, st, InputXf{}, addend, so);
    else launch_gemm_v<T, false, false>(x, wv, y, HW, tiles_per_img, total_tiles, st, InputXf{}, addend, so);
    return check_launch(name);
}

static int gemm_entry(const char* name, const void* x, WView wv, void* y, int64_t N, int64_t HW, int dtype,
                      void* stream, const void* addend = nullptr, StatOut so = StatOut{nullptr, 0}) {
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case OFASR_F32: return launch_gemm<float>(name, x, wv, y, N, HW, st, InputXf{}, addend, so);
        case OFASR_F16: return launch_gemm<f16_t>(name, x, wv, y, N, HW, st, InputXf{}, addend, so);
        default: return launch_gemm<bf16_t>(name, x, wv, y, N, HW, st, InputXf{}, addend, so);
    }
}

static int check_pw_args(const char* name, const void* a, const void* b, const void* c, int64_t ldw, int64_t N,
                         int64_t Cin, int64_t Cout, int64_t HW, int dtype) {
    OFASR_REQUIRE(a && b && c, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N >= 0 && Cin > 0 && Cout > 0 && HW >= 0, OFASR_ERR_INVALID_ARG,
                  "%s: bad shape N=%lld Cin=%lld Cout=%lld HW=%lld", name, (long long)N, (long long)Cin,
                  (long long)Cout, (long long)HW);
    OFASR_REQUIRE(ldw >= Cin, OFASR_ERR_INVALID_ARG, "%s: ldw=%lld < Cin=%lld", name, (long long)ldw, (long long)Cin);
    OFASR_REQUIRE(dtype == OFASR_F32 || dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_INVALID_ARG,
                  "%s: bad dtype %d", name, dtype);
    OFASR_REQUIRE(Cin <= 65536 && Cout <= 65536 && HW <= (1 << 30) && N <= (1 << 24), OFASR_ERR_UNSUPPORTED,
                  "%s: dimension too large", name);
    return OFASR_OK;
}

// XF: false plain; true: x is read through InputXf; BXM (with XF false): dy is formed from (dy = da, bx.y) through BwdXf
template <typename T, bool XF = false, bool BXM = false>
static int launch_wgrad(const char* name, const void* dy, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin,
                        int64_t Cout, int64_t HW, float* ws, hipStream_t st, InputXf xf = InputXf{}, BwdXf bx = BwdXf{}) {
    const WgradPlan p = wgrad_plan(N, Cin, Cout, HW, WgStage<T>::SUB);
    const bool big_is_dy = Cout >= Cin;
    const T* R = (const T*)(big_is_dy ? dy : x);
    const T* S = (const T*)(big_is_dy ? x : dy);
    // out(r, s): r indexes the big operand's channels
    const long long sr = big_is_dy ? ldw : 1, ss = big_is_dy ? 1 : ldw;
    const bool al = aligned_for(dy, x, HW, Elem<T>::is16);
    prof_note((double)sizeof(T) * (double)N * (double)HW * (double)(Cin + Cout) + 4.0 * (double)Cin * (double)Cout,
              2.0 * (double)N * (double)HW * (double)Cin * (double)Cout);
    if constexpr (Elem<T>::is16) {
        if (al) {
            const WdPlan wp = wd_plan(N, Cin, Cout, HW);
            dim3 grid((unsigned)wp.nsplit, (unsigned)cdiv(wp.MR, WD_ROWS), (unsigned)cdiv(wp.NS, WD_COLS));
            if constexpr (XF) {
                if (big_is_dy)   // x is the S operand
                    OFASR_LAUNCH((pw_wgrad_direct_kernel<T, 2>), grid, dim3(512), 0, st, R, S, ws, wp.MR, wp.NS,
                                       (int)HW, wp, xf);
                else
                    OFASR_LAUNCH((pw_wgrad_direct_kernel<T, 1>), grid, dim3(512), 0, st, R, S, ws, wp.MR, wp.NS,
                                       (int)HW, wp, xf);
            } else if constexpr (BXM) {
                if (!big_is_dy) {
                    set_error("%s: the fused BN backward is implemented for the wide gradient operand (Cout >= Cin)", name);
                    return OFASR_ERR_UNSUPPORTED;
                }
                OFASR_LAUNCH((pw_wgrad_direct_kernel<T, 3>), grid, dim3(512), 0, st, R, S, ws, wp.MR, wp.NS, (int)HW, wp,
                             InputXf{}, bx);
            } else {
                OFASR_LAUNCH((pw_wgrad_direct_kernel<T>), grid, dim3(512), 0, st, R, S, ws, wp.MR, wp.NS, (int)HW,
                                   wp);
            }
            int rc = check_launch(name);
            if (rc) return rc;
            const long long tot = (long long)wp.MR * wp.NS;
            OFASR_LAUNCH(pw_wgrad_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(64 * WR_ZL), 0, st, ws, dw, wp.MR,
                               wp.NS, wp.nsplit, sr, ss);
            return check_launch(name);
        }
    }
    if constexpr (XF || BXM) {
        set_error("%s: fused input transform needs the aligned 16-bit kernel", name);
        return OFASR_ERR_UNSUPPORTED;
    }
    if constexpr (!Elem<T>::is16) {
        // OFF by default (OFASR_PW_WGRAD_F32_DIRECT=1): 68 against 71.6 us per call in isolation, but the fp32 training step is
        // 22.96 against 22.78 ms with it (two A/B pairs) -- a wave-load of 64 lanes in 32 different rows runs at about one
        // lane per clock in the address path, which is what bounds this kernel, and it competes with the chain for that path
        static const bool f32_direct = [] { const char* e = getenv("OFASR_PW_WGRAD_F32_DIRECT"); return e && e[0] == '1'; }();
        if (al && f32_direct) {
            const WdPlan32 wp = wd_plan32(N, Cin, Cout, HW);
            dim3 grid((unsigned)wp.nsplit, (unsigned)cdiv(wp.MR, WD_ROWS), (unsigned)cdiv(wp.NS, WD_COLS));
            OFASR_LAUNCH(pw_wgrad_direct_f32_kernel, grid, dim3(512), 0, st, (const float*)R, (const float*)S, ws, wp.MR, wp.NS,
                         (int)HW, wp);
            int rc = check_launch(name);
            if (rc) return rc;
            const long long tot = (long long)wp.MR * wp.NS;
            OFASR_LAUNCH(pw_wgrad_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(64 * WR_ZL), 0, st, ws, dw, wp.MR, wp.NS,
                         wp.nsplit, sr, ss);
            return check_launch(name);
        }
    }
    dim3 grid((unsigned)cdiv(p.MR, 64), (unsigned)cdiv(p.NS, 64), (unsigned)p.nsplit);
    if (al)
        OFASR_LAUNCH((pw_wgrad_kernel<T, true>), grid, dim3(PW_THREADS), 0, st, R, S, ws, p.MR, p.NS, (int)HW,
                           p.stages_per_img, p.total_stages, p.stages_per_split);
    else
        OFASR_LAUNCH((pw_wgrad_kernel<T, false>), grid, dim3(PW_THREADS), 0, st, R, S, ws, p.MR, p.NS, (int)HW,
                           p.stages_per_img, p.total_stages, p.stages_per_split);
    int rc = check_launch(name);
    if (rc) return rc;
    const long long tot = (long long)p.MR * p.NS;
    OFASR_LAUNCH(pw_wgrad_reduce_kernel, dim3((unsigned)cdiv(tot, 64)), dim3(64 * WR_ZL), 0, st, ws, dw, p.MR, p.NS,
                       p.nsplit, sr, ss);
    return check_launch(name);
}

int pwconv_dgrad_add(const void* dy, const float* w, int64_t ldw, void* dx, const void* addend, int64_t N, int64_t Cin,
                     int64_t Cout, int64_t HW, int dtype, void* stream) {
    const char* name = "pwconv_dgrad_add";
    int rc = check_pw_args(name, dy, w, dx, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(addend != nullptr, OFASR_ERR_INVALID_ARG, "%s: null addend", name);
    if (N * HW == 0) return OFASR_OK;
    WView wv{w, 1, ldw, (int)Cin, (int)Cout};
    return gemm_entry(name, dy, wv, dx, N, HW, dtype, stream, addend);
}

// expand input gradient with the gradient operand dy1 read through the BN(+ReLU6) backward:  dx = W^T dy1(da, y) + addend.
// Only the pipelined fan-in kernel implements it: 16-bit, aligned, 64 < Cout <= FOLD_KMAX, vector-loadable weight slice.
bool pwconv_dgrad_bx_supported(const void* da, const void* y, const void* dx, const void* addend, const float* w,
                               int64_t ldw, int64_t Cin, int64_t Cout, int64_t HW, int dtype) {
    if (dtype != OFASR_F16 && dtype != OFASR_BF16) return false;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx) |
                           reinterpret_cast<uintptr_t>(addend) | reinterpret_cast<uintptr_t>(w);   // dy_out: checked by the caller
    static const bool fast_ok = [] { const char* e = getenv("OFASR_PW_FANIN_FAST"); return !(e && e[0] == '0'); }();
    return fast_ok && (bits & 15) == 0 && HW % 8 == 0 && ldw % 4 == 0 && Cin % 4 == 0 && Cout % 4 == 0 && Cout > 64 &&
           Cout <= FOLD_KMAX && Cout * HW < (1LL << 31) && (Cin + Cout) * ldw < (1LL << 31);   // 32-bit lane offsets
}

template <typename T>
static int launch_dgrad_bx(const char* name, const void* da, WView wv, void* dx, const void* addend, int64_t N, int64_t HW,
                           BwdXf bx, hipStream_t st) {
    const int tiles_per_img = (int)cdiv(HW, PW_TILE);
    const int64_t total64 = N * tiles_per_img;
    OFASR_REQUIRE(total64 <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "%s: too many pixel tiles", name);
    {
        const double px = (double)N * (double)HW, es = (double)sizeof(T);
        prof_note(es * px * (2.0 * wv.K + wv.M + (addend ? wv.M : 0)) + 4.0 * wv.K * wv.M, 2.0 * px * wv.K * wv.M);
    }
    dim3 grid((unsigned)total64, (unsigned)cdiv(wv.M, 64));
    OFASR_LAUNCH((pw_fanin_pipe_kernel<T, false, true, true>), grid, dim3(PW_THREADS), 0, st, (const T*)da, wv, (T*)dx,
                 (int)HW, tiles_per_img, (int)cdiv(wv.K, 64), InputXf{}, (const T*)addend, StatOut{nullptr, 0}, BnFold{}, bx);
    return check_launch(name);
}

int pwconv_dgrad_add_bx(const void* da, const float* w, int64_t ldw, void* dx, const void* addend, int64_t N, int64_t Cin,
                        int64_t Cout, int64_t HW, int dtype, BwdXf bx, void* stream) {
    const char* name = "pwconv_dgrad_add_bx";
    int rc = check_pw_args(name, da, w, dx, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(bx.y && bx.mean && bx.scale && bx.shift &&
                      ((bx.ka && bx.kbi) || (bx.fold_partial && bx.fold_invstd && bx.fold_P > 0 && bx.fold_C == Cout)),
                  OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(pwconv_dgrad_bx_supported(da, bx.y, dx, addend, w, ldw, Cin, Cout, HW, dtype), OFASR_ERR_UNSUPPORTED,
                  "%s: needs the pipelined fan-in kernel (aligned 16-bit tensors, 64 < Cout <= %d)", name, FOLD_KMAX);
    if (N * HW == 0) return OFASR_OK;
    WView wv{w, 1, ldw, (int)Cin, (int)Cout};   // dgrad: M = Cin rows of W^T, K = Cout
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16) return launch_dgrad_bx<f16_t>(name, da, wv, dx, addend, N, HW, bx, st);
    return launch_dgrad_bx<bf16_t>(name, da, wv, dx, addend, N, HW, bx, st);
}

bool pwconv_xf_supported(const void* x, const void* y, int64_t HW, int dtype) {
    return (dtype == OFASR_F16 || dtype == OFASR_BF16) && aligned_for(x, y, HW, true);
}

// project-conv input gradient da2 = W2^T dy3 that also leaves the BN2-backward sums of what it writes (BwdStatOut):
// the slab-walk kernel only (16-bit, aligned, vector weights, 64 gradient channels in, 256 / 384 out, HW % 128 == 0)
bool pwconv_dgrad_bstat_supported(const void* dy, const void* y, const void* dx, const float* w, int64_t ldw, int64_t Cin,
                                  int64_t Cout, int64_t HW, int dtype) {
    const bool wvec = (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (ldw % 4) == 0;
    return (dtype == OFASR_F16 || dtype == OFASR_BF16) && aligned_for(dy, dx, HW, true) &&
           (reinterpret_cast<uintptr_t>(y) & 15) == 0 && wvec && Cout == 64 && (Cin == 2 * FO_ROWS || Cin == 3 * FO_ROWS) &&
           HW % PW_TILE == 0;
}

template <typename T>
static int launch_dgrad_bstat(const void* dy, WView wv, void* dx, int64_t N, int64_t HW, BwdStatOut bs, hipStream_t st) {
    const int tiles_per_img = (int)(HW / PW_TILE);
    const int total_tiles = (int)(N * tiles_per_img);
    const double px = (double)N * (double)HW;
    prof_note(2.0 * px * (wv.K + 2.0 * wv.M) + 4.0 * wv.K * wv.M, 2.0 * px * wv.K * wv.M);
    if (wv.M == 3 * FO_ROWS)
        OFASR_LAUNCH((pw_fanout_slabs_kernel<T, 3, 2>), dim3((unsigned)total_tiles), dim3(PW_THREADS), 0, st, (const T*)dy, wv,
                     (T*)dx, (int)HW, tiles_per_img, StatOut{nullptr, 0}, bs);
    else
        OFASR_LAUNCH((pw_fanout_slabs_kernel<T, 2, 2>), dim3((unsigned)total_tiles), dim3(PW_THREADS), 0, st, (const T*)dy, wv,
                     (T*)dx, (int)HW, tiles_per_img, StatOut{nullptr, 0}, bs);
    return check_launch("pwconv_dgrad_bstat");
}

int pwconv_dgrad_bstat(const void* dy, const float* w, int64_t ldw, void* dx, int64_t N, int64_t Cin, int64_t Cout,
                       int64_t HW, int dtype, BwdStatOut bs, void* stream) {
    const char* name = "pwconv_dgrad_bstat";
    int rc = check_pw_args(name, dy, w, dx, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(bs.y && bs.mean && bs.scale && bs.shift && bs.partial, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(pwconv_dgrad_bstat_supported(dy, bs.y, dx, w, ldw, Cin, Cout, HW, dtype), OFASR_ERR_UNSUPPORTED,
                  "%s: shape outside the slab-walk kernel", name);
    OFASR_REQUIRE(N * (HW / PW_TILE) <= INT32_MAX && bs.P == (int)(N * (HW / PW_TILE)), OFASR_ERR_INVALID_ARG,
                  "%s: unit count %d does not match the launch", name, bs.P);
    WView wv{w, 1, ldw, (int)Cin, (int)Cout};  // rows = input channels, reduction over output channels
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16) return launch_dgrad_bstat<f16_t>(dy, wv, dx, N, HW, bs, st);
    return launch_dgrad_bstat<bf16_t>(dy, wv, dx, N, HW, bs, st);
}

int pwconv_stat_units(int64_t N, int64_t Cin, int64_t HW) {
    const int64_t tiles = N * cdiv(HW, PW_TILE);
    return (int)(Cin <= 64 ? tiles : 2 * tiles);
}

int pwconv_fwd_stat(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, StatOut so, void* stream) {
    const char* name = "pwconv_fwd_stat";
    int rc = check_pw_args(name, x, w, y, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(so.partial != nullptr && N * HW > 0, OFASR_ERR_INVALID_ARG, "%s: null statistics / empty tensor", name);
    WView wv{w, ldw, 1, (int)Cout, (int)Cin};
    return gemm_entry(name, x, wv, y, N, HW, dtype, stream, nullptr, so);
}

bool pwconv_fold_supported(const void* x, const void* y, const float* w, int64_t ldw, int64_t Cin, int64_t HW, int dtype) {
    const bool wvec = (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (ldw % 4) == 0;
    return pwconv_xf_supported(x, y, HW, dtype) && wvec && Cin > 64 && Cin <= FOLD_KMAX && Cin % 4 == 0;
}

int pwconv_fwd_fold(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, BnFold fold, void* stream, StatOut so) {
    const char* name = "pwconv_fwd_fold";
    int rc = check_pw_args(name, x, w, y, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(fold.cp && fold.P > 0 && fold.mean && fold.invstd && fold.scale && fold.shift, OFASR_ERR_INVALID_ARG,
                  "%s: incomplete statistics source", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    OFASR_REQUIRE(N * HW > 0, OFASR_ERR_UNSUPPORTED, "%s: empty tensor", name);
    WView wv{w, ldw, 1, (int)Cout, (int)Cin};
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16) return launch_gemm<f16_t, true>(name, x, wv, y, N, HW, st, InputXf{}, nullptr, so, fold);
    return launch_gemm<bf16_t, true>(name, x, wv, y, N, HW, st, InputXf{}, nullptr, so, fold);
}

int pwconv_fwd_xf(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t HW,
                  int dtype, InputXf xf, void* stream, StatOut so) {
    const char* name = "pwconv_fwd_xf";
    int rc = check_pw_args(name, x, w, y, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(xf.scale && xf.shift && xf.mean, OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    if (N * HW == 0) return OFASR_OK;
    WView wv{w, ldw, 1, (int)Cout, (int)Cin};
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16) return launch_gemm<f16_t, true>(name, x, wv, y, N, HW, st, xf, nullptr, so);
    return launch_gemm<bf16_t, true>(name, x, wv, y, N, HW, st, xf, nullptr, so);
}

int pwconv_wgrad_xf(const void* dy, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin, int64_t Cout,
                    int64_t HW, int dtype, InputXf xf, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "pwconv_wgrad_xf";
    int rc = check_pw_args(name, dy, x, dw, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(xf.scale && xf.shift && xf.mean, OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit only", name);
    OFASR_REQUIRE(N * HW > 0, OFASR_ERR_UNSUPPORTED, "%s: empty tensor", name);
    const size_t need = ofasr_pwconv_wgrad_workspace(N, Cin, Cout, HW);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16)
        return launch_wgrad<f16_t, true>(name, dy, x, dw, ldw, N, Cin, Cout, HW, (float*)workspace, st, xf);
    return launch_wgrad<bf16_t, true>(name, dy, x, dw, ldw, N, Cin, Cout, HW, (float*)workspace, st, xf);
}

bool pwconv_wgrad_bx_supported(const void* da, const void* y, const void* x, int64_t Cin, int64_t Cout, int64_t HW, int dtype) {
    if (!(dtype == OFASR_F16 || dtype == OFASR_BF16) || Cout < Cin) return false;
    return aligned_for(da, x, HW, true) && aligned_for(y, x, HW, true);
}

// dw[:Cout, :Cin] = sum dy * x with dy = BN(+ReLU6) backward of (da, bx.y), finished coefficients in bx.ka / bx.kbi
int pwconv_wgrad_bx(const void* da, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin, int64_t Cout, int64_t HW,
                    int dtype, BwdXf bx, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "pwconv_wgrad_bx";
    int rc = check_pw_args(name, da, x, dw, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(bx.y && bx.mean && bx.scale && bx.shift && bx.ka && bx.kbi, OFASR_ERR_INVALID_ARG, "%s: null transform", name);
    OFASR_REQUIRE(pwconv_wgrad_bx_supported(da, bx.y, x, Cin, Cout, HW, dtype), OFASR_ERR_UNSUPPORTED,
                  "%s: needs 16-bit aligned tensors and Cout >= Cin", name);
    OFASR_REQUIRE(N * HW > 0, OFASR_ERR_UNSUPPORTED, "%s: empty tensor", name);
    const size_t need = ofasr_pwconv_wgrad_workspace(N, Cin, Cout, HW);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    if (dtype == OFASR_F16)
        return launch_wgrad<f16_t, false, true>(name, da, x, dw, ldw, N, Cin, Cout, HW, (float*)workspace, st, InputXf{}, bx);
    return launch_wgrad<bf16_t, false, true>(name, da, x, dw, ldw, N, Cin, Cout, HW, (float*)workspace, st, InputXf{}, bx);
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT int ofasr_pwconv_fwd(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin,
                                  int64_t Cout, int64_t HW, int dtype, void* stream) {
    const char* name = "ofasr_pwconv_fwd";
    int rc = check_pw_args(name, x, w, y, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    if (N * HW == 0) return OFASR_OK;
    WView wv{w, ldw, 1, (int)Cout, (int)Cin};
    return gemm_entry(name, x, wv, y, N, HW, dtype, stream);
}

OFASR_EXPORT int ofasr_pwconv_dgrad(const void* dy, const float* w, int64_t ldw, void* dx, int64_t N, int64_t Cin,
                                    int64_t Cout, int64_t HW, int dtype, void* stream) {
    const char* name = "ofasr_pwconv_dgrad";
    int rc = check_pw_args(name, dy, w, dx, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    if (N * HW == 0) return OFASR_OK;
    WView wv{w, 1, ldw, (int)Cin, (int)Cout};  // rows = input channels, reduction over output channels
    return gemm_entry(name, dy, wv, dx, N, HW, dtype, stream);
}

OFASR_EXPORT size_t ofasr_pwconv_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t HW) {
    if (N <= 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return 0;
    // the 16-bit and fp32 kernels stage different pixel counts per barrier: size for the larger plan
    const WgradPlan p4 = wgrad_plan(N, Cin, Cout, HW, 4), p2 = wgrad_plan(N, Cin, Cout, HW, 2);
    int ns = p4.nsplit > p2.nsplit ? p4.nsplit : p2.nsplit;
    const int nd = wd_plan(N, Cin, Cout, HW).nsplit;
    ns = nd > ns ? nd : ns;
    const int nd32 = wd_plan32(N, Cin, Cout, HW).nsplit;
    ns = nd32 > ns ? nd32 : ns;
    return (size_t)ns * (size_t)p4.MR * (size_t)p4.NS * sizeof(float);
}

OFASR_EXPORT int ofasr_pwconv_wgrad(const void* dy, const void* x, float* dw, int64_t ldw, int64_t N, int64_t Cin,
                                    int64_t Cout, int64_t HW, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    const char* name = "ofasr_pwconv_wgrad";
    int rc = check_pw_args(name, dy, x, dw, ldw, N, Cin, Cout, HW, dtype);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (N * HW == 0) {
        // empty reduction: the slice gradient is zero
        for (int64_t co = 0; co < Cout; ++co) {
            hipError_t e = hipMemsetAsync(dw + co * ldw, 0, (size_t)Cin * sizeof(float), st);
            OFASR_REQUIRE(e == hipSuccess, OFASR_ERR_LAUNCH, "%s: memset failed", name);
        }
        return OFASR_OK;
    }
    const size_t need = ofasr_pwconv_wgrad_workspace(N, Cin, Cout, HW);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, need);
    float* ws = (float*)workspace;
    switch (dtype) {
        case OFASR_F32: return launch_wgrad<float>(name, dy, x, dw, ldw, N, Cin, Cout, HW, ws, st);
        case OFASR_F16: return launch_wgrad<f16_t>(name, dy, x, dw, ldw, N, Cin, Cout, HW, ws, st);
        default: return launch_wgrad<bf16_t>(name, dy, x, dw, ldw, N, Cin, Cout, HW, ws, st);
    }
}

