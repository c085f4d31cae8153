// pixel_shuffle.hip -- sub-pixel reshuffle (PixelShuffle / PixelUnshuffle) for gfx950.
//
// Replaces nn.PixelShuffle(2) (reference ofa/utils.py:309-310; ConvLayer act 'pixelshuffle',
// ofa_mbs4.py:120) and the one-hot strided conv of pixel_unshuffle (ofa/utils.py:383-397).
//
// Pure HBM-bound byte permutation: algorithmic bytes = 2 * N*C*r*r*H*W*elem_size (each byte read
// once, written once).  Elements are moved as integers => bit-exact for every dtype.
//
// r == 2 fast path: one thread owns a 16-byte run of one LR row.  It loads that run from the 4
// source planes (i,j in {0,1}) -- 16 B per lane, lanes consecutive within a plane, so a wave reads
// four contiguous 1 KiB spans -- interleaves in registers and writes two 32-byte runs of the two
// HR rows (2h, 2h+1).  Unshuffle runs the same map backwards.
#include "ofasr_common.h"

namespace ofasr {

// ------------------------------------------------------------------------------------ generic
template <typename U, bool SHUFFLE>
__global__ void __launch_bounds__(256) ps_generic_kernel(const U* __restrict__ x, U* __restrict__ y,
                                                         int64_t total, int C, int H, int W, int r) {
    // index space: the HR tensor [N, C, H*r, W*r]
    const int Ho = H * r, Wo = W * r;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        int wo = (int)(idx % Wo);
        int64_t t = idx / Wo;
        int ho = (int)(t % Ho);
        t /= Ho;
        int c = (int)(t % C);
        int64_t n = t / C;
        int h = ho / r, i = ho % r, w = wo / r, j = wo % r;
        int64_t lr = (((n * C + c) * r * r + i * r + j) * H + h) * (int64_t)W + w;
        if (SHUFFLE) y[idx] = x[lr];
        else y[lr] = x[idx];
    }
}

// ------------------------------------------------------------------------------- r == 2, 16 B
template <int ES> struct Interleave;
template <> struct Interleave<4> {
    // a = 4 elements of plane j=0, b = plane j=1 -> 8 interleaved elements
    static __device__ __forceinline__ void zip(const uint4& a, const uint4& b, uint4& lo, uint4& hi) {
        lo = make_uint4(a.x, b.x, a.y, b.y);
        hi = make_uint4(a.z, b.z, a.w, b.w);
    }
    static __device__ __forceinline__ void unzip(const uint4& lo, const uint4& hi, uint4& a, uint4& b) {
        a = make_uint4(lo.x, lo.z, hi.x, hi.z);
        b = make_uint4(lo.y, lo.w, hi.y, hi.w);
    }
};
template <> struct Interleave<2> {
    static __device__ __forceinline__ uint32_t zlo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
    static __device__ __forceinline__ uint32_t zhi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }
    static __device__ __forceinline__ void zip(const uint4& a, const uint4& b, uint4& lo, uint4& hi) {
        lo = make_uint4(zlo(a.x, b.x), zhi(a.x, b.x), zlo(a.y, b.y), zhi(a.y, b.y));
        hi = make_uint4(zlo(a.z, b.z), zhi(a.z, b.z), zlo(a.w, b.w), zhi(a.w, b.w));
    }
    // inverse of zip: dwords (p, q) = (zlo, zhi) -> a = zlo(p,q), b = zhi(p,q)
    static __device__ __forceinline__ void unzip(const uint4& lo, const uint4& hi, uint4& a, uint4& b) {
        a = make_uint4(zlo(lo.x, lo.y), zlo(lo.z, lo.w), zlo(hi.x, hi.y), zlo(hi.z, hi.w));
        b = make_uint4(zhi(lo.x, lo.y), zhi(lo.z, lo.w), zhi(hi.x, hi.y), zhi(hi.z, hi.w));
    }
};

// items: (nc, h, wq) with wq indexing 16-byte runs of an LR row; Wq = W*ES/16 runs per row.
template <int ES, bool SHUFFLE>
__global__ void __launch_bounds__(256) ps_r2_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                    int64_t items, int H, int Wq) {
    const int64_t plane = (int64_t)H * Wq;  // uint4 per LR plane
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items;
         it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = it / plane;
        const int64_t rem = it - nc * plane;  // h*Wq + wq
        const int h = (int)(rem / Wq);
        const int wq = (int)(rem - (int64_t)h * Wq);
        // LR side: planes nc*4 + {0,1,2,3}; HR side: plane nc, rows 2h / 2h+1, runs 2wq / 2wq+1
        const int64_t lr0 = nc * 4 * plane + rem;
        const int64_t hr0 = nc * 4 * plane + ((int64_t)(2 * h) * (2 * Wq)) + 2 * wq;
        if (SHUFFLE) {
            uint4 a0 = src[lr0], b0 = src[lr0 + plane], a1 = src[lr0 + 2 * plane], b1 = src[lr0 + 3 * plane];
            uint4 lo, hi;
            Interleave<ES>::zip(a0, b0, lo, hi);
            dst[hr0] = lo;
            dst[hr0 + 1] = hi;
            Interleave<ES>::zip(a1, b1, lo, hi);
            dst[hr0 + 2 * Wq] = lo;
            dst[hr0 + 2 * Wq + 1] = hi;
        } else {
            uint4 lo0 = src[hr0], hi0 = src[hr0 + 1], lo1 = src[hr0 + 2 * Wq], hi1 = src[hr0 + 2 * Wq + 1];
            uint4 a, b;
            Interleave<ES>::unzip(lo0, hi0, a, b);
            dst[lr0] = a;
            dst[lr0 + plane] = b;
            Interleave<ES>::unzip(lo1, hi1, a, b);
            dst[lr0 + 2 * plane] = a;
            dst[lr0 + 3 * plane] = b;
        }
    }
}

// BatchNorm apply + PixelShuffle(2) in one pass (ConvLayer with act_func "pixelshuffle": conv -> BN -> PixelShuffle,
// reference ofa/layers.py:120-151, ofa_mbs4.py:111-123): a thread loads the same 8 pixels of the 4 source channels of one
// output channel, applies each channel's (v - mean) * scale + beta (bn_act_fwd_kernel's arithmetic: beta = shift +
// mean * scale) and writes two 32-byte runs of the up-sampled plane.  16-bit tensors.
template <typename T>
__device__ __forceinline__ uint4 ps_bn8(uint4 v, float mu, float sc, float be) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float lo, hi;
        unpack2<T>(w[i], lo, hi);
        o[i] = pack2<T>(fmaf(lo - mu, sc, be), fmaf(hi - mu, sc, be));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

template <typename T>
__global__ void __launch_bounds__(256) ps_r2_bn_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                       const float* __restrict__ mean, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int64_t items, int Cout, int H,
                                                       int Wq) {
    const int64_t plane = (int64_t)H * Wq;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = it / plane;
        const int64_t rem = it - nc * plane;
        const int h = (int)(rem / Wq);
        const int wq = (int)(rem - (int64_t)h * Wq);
        const int c0 = 4 * (int)(nc % Cout);
        const int64_t lr0 = nc * 4 * plane + rem;
        const int64_t hr0 = nc * 4 * plane + ((int64_t)(2 * h) * (2 * Wq)) + 2 * wq;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = src[lr0 + j * plane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mu = mean[c0 + j], sc = scale[c0 + j];
            v[j] = ps_bn8<T>(v[j], mu, sc, fmaf(mu, sc, shift[c0 + j]));
        }
        uint4 lo, hi;
        Interleave<2>::zip(v[0], v[1], lo, hi);
        dst[hr0] = lo;
        dst[hr0 + 1] = hi;
        Interleave<2>::zip(v[2], v[3], lo, hi);
        dst[hr0 + 2 * Wq] = lo;
        dst[hr0 + 2 * Wq + 1] = hi;
    }
}

template <bool SHUFFLE>
static int launch(const void* x, void* y, int64_t N, int64_t C, int64_t H, int64_t W, int r, int es,
                  hipStream_t st) {
    const char* name = SHUFFLE ? "ofasr_pixel_shuffle" : "ofasr_pixel_unshuffle";
    OFASR_REQUIRE(x && y, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N >= 0 && C >= 0 && H >= 0 && W >= 0 && r >= 1, OFASR_ERR_INVALID_ARG,
                  "%s: bad shape N=%lld C=%lld H=%lld W=%lld r=%d", name, (long long)N, (long long)C,
                  (long long)H, (long long)W, r);
    OFASR_REQUIRE(es == 1 || es == 2 || es == 4 || es == 8, OFASR_ERR_UNSUPPORTED,
                  "%s: elem_size %d not in {1,2,4,8}", name, es);
    OFASR_REQUIRE(C <= INT32_MAX && H * r <= INT32_MAX && W * r <= INT32_MAX, OFASR_ERR_UNSUPPORTED,
                  "%s: dimension too large", name);
    const int64_t total = N * C * r * r * H * W;
    if (total == 0) return OFASR_OK;
    prof_note(2.0 * (double)total * es, 0.0);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    if (r == 2 && (es == 2 || es == 4) && aligned && (W * es) % 16 == 0) {
        const int Wq = (int)(W * es / 16);
        const int64_t items = N * C * H * Wq;
        const int64_t blocks = cdiv(items, 256);
        const int grid = (int)(blocks < 65536 ? blocks : 65536);
        if (es == 4)
            OFASR_LAUNCH((ps_r2_kernel<4, SHUFFLE>), dim3(grid), dim3(256), 0, st, (const uint4*)x,
                               (uint4*)y, items, (int)H, Wq);
        else
            OFASR_LAUNCH((ps_r2_kernel<2, SHUFFLE>), dim3(grid), dim3(256), 0, st, (const uint4*)x,
                               (uint4*)y, items, (int)H, Wq);
        return check_launch(name);
    }
    const int64_t blocks = cdiv(total, 256);
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
#define OFASR_PS_GENERIC(ES)                                                                       \
    OFASR_LAUNCH((ps_generic_kernel<typename uint_of<ES>::type, SHUFFLE>), dim3(grid), dim3(256), \
                       0, st, (const typename uint_of<ES>::type*)x, (typename uint_of<ES>::type*)y,    \
                       total, (int)C, (int)H, (int)W, r)
    switch (es) {
        case 1: OFASR_PS_GENERIC(1); break;
        case 2: OFASR_PS_GENERIC(2); break;
        case 4: OFASR_PS_GENERIC(4); break;
        default: OFASR_PS_GENERIC(8); break;
    }
#undef OFASR_PS_GENERIC
    return check_launch(name);
}

}  // namespace ofasr

OFASR_EXPORT int ofasr_pixel_shuffle(const void* x, void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                                     int r, int elem_size, void* stream) {
    return ofasr::launch<true>(x, y, N, C, H, W, r, elem_size, ofasr::as_stream(stream));
}

OFASR_EXPORT int ofasr_pixel_unshuffle(const void* x, void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                                       int r, int elem_size, void* stream) {
    return ofasr::launch<false>(x, y, N, C, H, W, r, elem_size, ofasr::as_stream(stream));
}

// y[N, C, 2H, 2W] = PixelShuffle(2)(BN(x[N, 4C, H, W])) with finalized statistics (stats = mean | invstd | scale | shift,
// 4*C_in floats as ofasr_bn_finalize_cp / ofasr_bn_fwd leave them); f16 / bf16, W % 8 == 0, 16-byte aligned tensors
OFASR_EXPORT int ofasr_pixel_shuffle2_bn(const void* x, void* y, const float* stats, int64_t N, int64_t C, int64_t H, int64_t W,
                                         int dtype, void* stream) {
    using namespace ofasr;
    const char* name = "ofasr_pixel_shuffle2_bn";
    OFASR_REQUIRE(x && y && stats, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit tensors only", name);
    OFASR_REQUIRE(W % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: needs W %% 8 == 0 and 16-byte aligned tensors", name);
    OFASR_REQUIRE(C <= INT32_MAX / 4 && 2 * H <= INT32_MAX && 2 * W <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    const int64_t Cin = 4 * C;
    const float *mean = stats, *scale = stats + 2 * Cin, *shift = stats + 3 * Cin;
    const int Wq = (int)(W / 8);
    const int64_t items = N * C * H * Wq;
    const int64_t blocks = cdiv(items, 256);
    const int grid = (int)(blocks < 65536 ? blocks : 65536);
    hipStream_t st = as_stream(stream);
    prof_note(2.0 * 2.0 * (double)N * (double)Cin * (double)H * (double)W, 0.0);
    if (dtype == OFASR_BF16)
        OFASR_LAUNCH((ps_r2_bn_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const uint4*)x, (uint4*)y, mean, scale, shift, items,
                     (int)C, (int)H, Wq);
    else
        OFASR_LAUNCH((ps_r2_bn_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const uint4*)x, (uint4*)y, mean, scale, shift, items,
                     (int)C, (int)H, Wq);
    return check_launch(name);
}
