// bnact.hip -- sliced BatchNorm2d fused with its activation (ReLU6) and the block's residual add.
//
// Replaces, per call site of the MB block (reference ofa/elastic_nn/modules/dynamic_layers.py:35-63:
// DynamicBatchNorm2d -> ReLU6, and proxyless_nets.py:50: + shortcut), the ATen chain
//     F.batch_norm (dynamic_op.py:163-167)  ->  hardtanh_  ->  add
// which on the GPU is 3-4 full passes over the mid tensor forward and 4-5 backward, by
//     forward : one read-only statistics pass + one fused normalise/activate/add pass
//     backward: one read-only reduction pass (sum dz, sum dz*xhat; dz = dy masked by the ReLU6 window,
//               recomputed from the PRE-BN tensor) + one fused apply pass.
// All four are pure HBM streaming kernels (roofline: HBM; algorithmic bytes are listed per kernel).
// Statistics are reduced through per-block partial slabs in a fixed order => deterministic, no atomics.
//
//   bn_stats        reads B*M*C                          (M = N*HW elements per channel, B = elem size)
//   bn_finalize     O(C)
//   bn_act_fwd      reads B*M*C (+B*M*C residual), writes B*M*C
//   bn_bwd_reduce   reads 2*B*M*C
//   bn_bwd_apply    reads 2*B*M*C, writes B*M*C
#include <stdlib.h>
#include "ofasr_common.h"

namespace ofasr {

constexpr int BN_THREADS = 256;

template <typename T> struct Vec;   // 16-byte vector of T
template <> struct Vec<float> { static constexpr int N = 4; };
template <> struct Vec<bf16_t> { static constexpr int N = 8; };
template <> struct Vec<f16_t> { static constexpr int N = 8; };

template <typename T>
__device__ __forceinline__ void unpack16(const uint4& v, float* out) {
    if constexpr (sizeof(T) == 4) {
        out[0] = __uint_as_float(v.x); out[1] = __uint_as_float(v.y);
        out[2] = __uint_as_float(v.z); out[3] = __uint_as_float(v.w);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T lo, hi;
            lo.v = (uint16_t)(w[i] & 0xffffu);
            hi.v = (uint16_t)(w[i] >> 16);
            out[2 * i] = to_float(lo);
            out[2 * i + 1] = to_float(hi);
        }
    }
}

template <typename T>
__device__ __forceinline__ uint4 pack16(const float* in) {
    if constexpr (sizeof(T) == 4) {
        return make_uint4(__float_as_uint(in[0]), __float_as_uint(in[1]), __float_as_uint(in[2]), __float_as_uint(in[3]));
    } else {
        return make_uint4(pack2<T>(in[0], in[1]), pack2<T>(in[2], in[3]), pack2<T>(in[4], in[5]), pack2<T>(in[6], in[7]));
    }
}

// wave-wide sum of a double (two 32-bit shuffles per step); result valid in every lane
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int lo = __shfl_xor(__double2loint(v), o, 64);
        const int hi = __shfl_xor(__double2hiint(v), o, 64);
        v += __hiloint2double(hi, lo);
    }
    return v;
}

// deterministic block reduction of two running sums (fp64: the variance is E[x^2] - mean^2, and the backward
// coefficients are differences of large sums -- fp32 partial sums cost ~1e-5 relative there); valid in thread 0
__device__ __forceinline__ void block_reduce2(double& a, double& b) {
    __shared__ double sa[BN_THREADS / 64], sb[BN_THREADS / 64];
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sa[w] = a;
        sb[w] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ra = 0.0, rb = 0.0;
#pragma unroll
        for (int i = 0; i < BN_THREADS / 64; ++i) {
            ra += sa[i];
            rb += sb[i];
        }
        a = ra;
        b = rb;
    }
    __syncthreads();
}

// work split: grid = (C, P); block (c, p) covers images n = p, p+P, ... of channel c.
// ----------------------------------------------------------------------------------- bn_stats
template <typename T, bool VEC>
__global__ void __launch_bounds__(BN_THREADS) bn_stats_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                              int N, int C, int HW, int P) {
    const int c = blockIdx.x, p = blockIdx.y;
    double s = 0.0, ss = 0.0;
    for (int n = p; n < N; n += P) {
        const T* xp = x + ((long long)n * C + c) * HW;
        if (VEC) {
            constexpr int V = Vec<T>::N;
            const uint4* xv = reinterpret_cast<const uint4*>(xp);
            for (int i = threadIdx.x; i < HW / V; i += BN_THREADS) {
                float f[8];
                unpack16<T>(xv[i], f);
                float s8 = 0.f, q8 = 0.f;   // one vector's worth in fp32, then into the fp64 running sums
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    s8 += f[j];
                    q8 = fmaf(f[j], f[j], q8);
                }
                s += (double)s8;
                ss += (double)q8;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += BN_THREADS) {
                const float v = to_float(xp[i]);
                s += (double)v;
                ss += (double)v * (double)v;
            }
        }
    }
    block_reduce2(s, ss);
    if (threadIdx.x == 0) {
        partial[((long long)p * C + c) * 2] = s;
        partial[((long long)p * C + c) * 2 + 1] = ss;
    }
}

// -------------------------------------------------------------------------------- bn_finalize
// mode 0 (train): mean / biased var from the partials; running stats EMA with the unbiased var.
// mode 1 (eval) : statistics are the running buffers.
// outputs (length C): mean, invstd, scale = gamma*invstd, shift = beta - mean*scale
__global__ void __launch_bounds__(64) bn_finalize_kernel(const double* __restrict__ partial, int P, int C, double M,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, double momentum, double eps,
                                                         int training, float* __restrict__ mean_out,
                                                         float* __restrict__ invstd_out, float* __restrict__ scale,
                                                         float* __restrict__ shift, int64_t* k0 = nullptr,
                                                         int64_t* k1 = nullptr, int64_t* k2 = nullptr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // num_batches_tracked of the composite block's BNs ride along
        if (k0) *k0 += 1;
        if (k1) *k1 += 1;
        if (k2) *k2 += 1;
    }
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (training) {
        double s = 0.0, ss = 0.0;
        for (int p = 0; p < P; ++p) {
            s += partial[((long long)p * C + c) * 2];
            ss += partial[((long long)p * C + c) * 2 + 1];
        }
        mean = s / M;
        var = ss / M - mean * mean;
        if (var < 0.0) var = 0.0;
        if (running_mean) {
            const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
        }
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    const double invstd = 1.0 / sqrt(var + eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    scale[c] = (float)(g * invstd);
    shift[c] = (float)(b - mean * g * invstd);
}

// --------------------------------------------------------------------------------- bn_act_fwd
// y = act(x*scale[c] + shift[c] (+ res));  act: 0 none, 1 relu6.   grid = (C, N-chunks)
// where the per-channel statistics of a forward apply pass come from
struct BnSource {
    int mode;                 // 0: given (scale, shift, mean);  1: train, from stats partials;  2: eval, running stats;
                              // 3: train, from the conv kernels' epilogue partials (StatOut: float2 [C][Pstat])
    const float2* cp;         // mode 3
    const double* partial;    // mode 1: [Pstat][C][2]
    int Pstat;
    double M, momentum, eps;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* mean;              // mode 1/2: outputs (written by the p == 0 block of each channel); mode 0: inputs
    float* invstd;
    float* scale;
    float* shift;
};

template <typename T, bool VEC, int ACT, bool RES>
__global__ void __launch_bounds__(BN_THREADS) bn_act_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                                T* __restrict__ y, BnSource src, int N, int C, int HW,
                                                                int P) {
    const int c = blockIdx.x, p = blockIdx.y;
    float sc, mu, sh;   // centred form (x - mean)*scale + beta: x*scale + shift cancels when |mean| >> std
    if (src.mode == 0) {
        sc = src.scale[c];
        mu = src.mean[c];
        sh = fmaf(mu, sc, src.shift[c]);
    } else {
        // every block folds the (tiny) finalize into itself: no separate launch between the passes
        double mean, var;
        if (src.mode == 3) {
            // up to ~1000 partials per channel: the block sums them cooperatively (fixed order), then broadcasts
            __shared__ double bc[2];
            double s = 0.0, ss = 0.0;
            for (int q = threadIdx.x; q < src.Pstat; q += BN_THREADS) {
                const float2 v = src.cp[(long long)c * src.Pstat + q];
                s += (double)v.x;
                ss += (double)v.y;
            }
            block_reduce2(s, ss);
            if (threadIdx.x == 0) {
                bc[0] = s;
                bc[1] = ss;
            }
            __syncthreads();
            mean = bc[0] / src.M;
            var = bc[1] / src.M - mean * mean;
            if (var < 0.0) var = 0.0;
        } else if (src.mode == 1) {
            double s = 0.0, ss = 0.0;
            __shared__ double ps[2][BN_THREADS];   // one request per thread, in-order sum from LDS (see bn_bwd_apply_kernel)
            if (src.Pstat <= BN_THREADS) {
                if ((int)threadIdx.x < src.Pstat) {
                    ps[0][threadIdx.x] = src.partial[((long long)threadIdx.x * C + c) * 2];
                    ps[1][threadIdx.x] = src.partial[((long long)threadIdx.x * C + c) * 2 + 1];
                }
                __syncthreads();
                for (int q = 0; q < src.Pstat; ++q) {
                    s += ps[0][q];
                    ss += ps[1][q];
                }
            } else {
                for (int q = 0; q < src.Pstat; ++q) {
                    s += src.partial[((long long)q * C + c) * 2];
                    ss += src.partial[((long long)q * C + c) * 2 + 1];
                }
            }
            mean = s / src.M;
            var = ss / src.M - mean * mean;
            if (var < 0.0) var = 0.0;
        } else {
            mean = (double)src.running_mean[c];
            var = (double)src.running_var[c];
        }
        const double invstd = 1.0 / sqrt(var + src.eps);
        const double g = src.gamma ? (double)src.gamma[c] : 1.0, b = src.beta ? (double)src.beta[c] : 0.0;
        sc = (float)(g * invstd);
        mu = (float)mean;
        sh = (float)b;
        if (p == 0 && threadIdx.x == 0) {
            src.mean[c] = (float)mean;
            src.invstd[c] = (float)invstd;
            src.scale[c] = sc;
            src.shift[c] = (float)(b - mean * g * invstd);
            if ((src.mode == 1 || src.mode == 3) && src.running_mean) {
                const double unb = src.M > 1.0 ? var * src.M / (src.M - 1.0) : var;
                src.running_mean[c] = (float)((1.0 - src.momentum) * (double)src.running_mean[c] + src.momentum * mean);
                src.running_var[c] = (float)((1.0 - src.momentum) * (double)src.running_var[c] + src.momentum * unb);
            }
        }
    }
    for (int n = p; n < N; n += P) {
        const long long off = ((long long)n * C + c) * HW;
        if (VEC) {
            constexpr int V = Vec<T>::N;
            const uint4* xv = reinterpret_cast<const uint4*>(x + off);
            const uint4* rv = RES ? reinterpret_cast<const uint4*>(res + off) : nullptr;
            uint4* yv = reinterpret_cast<uint4*>(y + off);
            for (int i = threadIdx.x; i < HW / V; i += BN_THREADS) {
                float f[8], r[8];
                unpack16<T>(xv[i], f);
                if (RES) unpack16<T>(rv[i], r);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float v = fmaf(f[j] - mu, sc, sh);
                    if (RES) v += r[j];
                    if (ACT == 1) v = fminf(fmaxf(v, 0.f), 6.f);
                    f[j] = v;
                }
                yv[i] = pack16<T>(f);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += BN_THREADS) {
                float v = fmaf(to_float(x[off + i]) - mu, sc, sh);
                if (RES) v += to_float(res[off + i]);
                if (ACT == 1) v = fminf(fmaxf(v, 0.f), 6.f);
                y[off + i] = from_float<T>(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------ bn_bwd_reduce
// dz = dy * [0 < x*scale+shift (+res) < 6]  (ACT == 1; the window is evaluated on the fp32 pre-activation,
// hardtanh_backward semantics);  partial[p][c] = (sum dz, sum dz * xhat),  xhat = (x - mean) * invstd
template <typename T, bool VEC, int ACT, bool RES>
__global__ void __launch_bounds__(BN_THREADS) bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const T* __restrict__ res,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd,
                                                                   double* __restrict__ partial, int N, int C, int HW,
                                                                   int P) {
    const int c = blockIdx.x, p = blockIdx.y;
    const float sc = scale[c], mu = mean[c], is = invstd[c], sh = fmaf(mu, sc, shift[c]);
    double s = 0.0, sx = 0.0;
    for (int n = p; n < N; n += P) {
        const long long off = ((long long)n * C + c) * HW;
        if (VEC) {
            constexpr int V = Vec<T>::N;
            const uint4* xv = reinterpret_cast<const uint4*>(x + off);
            const uint4* dv = reinterpret_cast<const uint4*>(dy + off);
            const uint4* rv = RES ? reinterpret_cast<const uint4*>(res + off) : nullptr;
            for (int i = threadIdx.x; i < HW / V; i += BN_THREADS) {
                float f[8], g[8], r[8];
                unpack16<T>(xv[i], f);
                unpack16<T>(dv[i], g);
                if (RES) unpack16<T>(rv[i], r);
                float s8 = 0.f, q8 = 0.f;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float dz = g[j];
                    if (ACT == 1) {
                        float pre = fmaf(f[j] - mu, sc, sh);
                        if (RES) pre += r[j];
                        dz = (pre > 0.f && pre < 6.f) ? dz : 0.f;
                    }
                    s8 += dz;
                    q8 = fmaf(dz, (f[j] - mu) * is, q8);
                }
                s += (double)s8;
                sx += (double)q8;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += BN_THREADS) {
                const float xf = to_float(x[off + i]);
                float dz = to_float(dy[off + i]);
                if (ACT == 1) {
                    float pre = fmaf(xf - mu, sc, sh);
                    if (RES) pre += to_float(res[off + i]);
                    dz = (pre > 0.f && pre < 6.f) ? dz : 0.f;
                }
                s += (double)dz;
                sx += (double)dz * (double)((xf - mu) * is);
            }
        }
    }
    block_reduce2(s, sx);
    if (threadIdx.x == 0) {
        partial[((long long)p * C + c) * 2] = s;
        partial[((long long)p * C + c) * 2 + 1] = sx;
    }
}

// dx = scale * (dz - a - xhat*b), a = mean(dz), b = mean(dz*xhat) folded from the reduction partials by every block
// (a = b = 0 in eval mode); the p == 0 block of each channel also writes dgamma / dbeta.  Optionally dres = dz.
template <typename T, bool VEC, int ACT, bool RES>
__global__ void __launch_bounds__(BN_THREADS) bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                  const T* __restrict__ res, T* __restrict__ dx,
                                                                  T* __restrict__ dres, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const double* __restrict__ partial, int Pred, double M,
                                                                  int training, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, int N, int C, int HW,
                                                                  int P) {
    const int c = blockIdx.x, p = blockIdx.y;
    const float sc = scale[c], mu = mean[c], is = invstd[c], sh = fmaf(mu, sc, shift[c]);
    double s = 0.0, sx = 0.0;
    // the channel's Pred partial pairs: one request per thread and an in-order sum from LDS (same order, same result as
    // the serial loop, which costs Pred dependent L2 round trips before the block's first tensor request)
    __shared__ double ps[2][BN_THREADS];
    if (Pred <= BN_THREADS) {
        if ((int)threadIdx.x < Pred) {
            ps[0][threadIdx.x] = partial[((long long)threadIdx.x * C + c) * 2];
            ps[1][threadIdx.x] = partial[((long long)threadIdx.x * C + c) * 2 + 1];
        }
        __syncthreads();
        for (int q = 0; q < Pred; ++q) {
            s += ps[0][q];
            sx += ps[1][q];
        }
    } else {
        for (int q = 0; q < Pred; ++q) {
            s += partial[((long long)q * C + c) * 2];
            sx += partial[((long long)q * C + c) * 2 + 1];
        }
    }
    if (p == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = (float)sx;
        if (dbeta) dbeta[c] = (float)s;
    }
    const float k1 = sc, ka = training ? (float)(s / M) : 0.f, kb = training ? (float)(sx / M) : 0.f;
    for (int n = p; n < N; n += P) {
        const long long off = ((long long)n * C + c) * HW;
        if (VEC) {
            constexpr int V = Vec<T>::N;
            const uint4* xv = reinterpret_cast<const uint4*>(x + off);
            const uint4* dv = reinterpret_cast<const uint4*>(dy + off);
            const uint4* rv = RES ? reinterpret_cast<const uint4*>(res + off) : nullptr;
            uint4* ov = reinterpret_cast<uint4*>(dx + off);
            uint4* orv = (RES && dres) ? reinterpret_cast<uint4*>(dres + off) : nullptr;
            for (int i = threadIdx.x; i < HW / V; i += BN_THREADS) {
                float f[8], g[8], r[8], o[8];
                unpack16<T>(xv[i], f);
                unpack16<T>(dv[i], g);
                if (RES) unpack16<T>(rv[i], r);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float dz = g[j];
                    if (ACT == 1) {
                        float pre = fmaf(f[j] - mu, sc, sh);
                        if (RES) pre += r[j];
                        dz = (pre > 0.f && pre < 6.f) ? dz : 0.f;
                    }
                    g[j] = dz;
                    o[j] = k1 * (dz - ka - (f[j] - mu) * is * kb);
                }
                ov[i] = pack16<T>(o);
                if (orv) orv[i] = pack16<T>(g);
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += BN_THREADS) {
                const float xf = to_float(x[off + i]);
                float dz = to_float(dy[off + i]);
                if (ACT == 1) {
                    float pre = fmaf(xf - mu, sc, sh);
                    if (RES) pre += to_float(res[off + i]);
                    dz = (pre > 0.f && pre < 6.f) ? dz : 0.f;
                }
                dx[off + i] = from_float<T>(k1 * (dz - ka - (xf - mu) * is * kb));
                if (RES && dres) dres[off + i] = from_float<T>(dz);
            }
        }
    }
}


// ---- one-pass BatchNorm backward for SMALL channels (round 3): a channel whose N * HW elements fit the registers of one
// 1024-thread workgroup (<= 65536 16-bit values: BN3 of every MB block and the 64-channel static convs at LR resolution,
// 8 MB tensors) is reduced and applied by the SAME workgroup -- each thread keeps its <= 8 chunks of dy and x in registers
// across the block-wide sum.  One launch instead of two and one pass over (dy, x) instead of two; on the input-gradient
// chain every kernel boundary is a pipeline bubble, which is what the two 7 us kernels mostly were.  No activation, no
// residual (the shapes above).  Sums: fp32 per thread (<= 64 values), fp64 across the workgroup in a fixed order.
constexpr int BN1P_THREADS = 1024, BN1P_MAXCH = 8;
template <typename T>
__global__ void __launch_bounds__(BN1P_THREADS) bn_bwd_onepass_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                      T* __restrict__ dx, const float* __restrict__ scale,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      int N, int C, int HW, double M, int training) {
    __shared__ double red[2][BN1P_THREADS / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int cpi = HW / 8;                 // 16-byte chunks per image plane
    const int total = N * cpi;
    const float mu = mean[c], is = invstd[c], k1 = scale[c];
    uint4 dv[BN1P_MAXCH], xv[BN1P_MAXCH];
    long long off[BN1P_MAXCH];
#pragma unroll
    for (int j = 0; j < BN1P_MAXCH; ++j) {
        const int q = tid + j * BN1P_THREADS;
        const int qc = q < total ? q : total - 1;        // (clamped request, zeroed below: no load under a branch)
        const int n = qc / cpi, i = qc - n * cpi;
        off[j] = ((long long)n * C + c) * HW + 8LL * i;
        dv[j] = *reinterpret_cast<const uint4*>(dy + off[j]);
        xv[j] = *reinterpret_cast<const uint4*>(x + off[j]);
    }
    float s = 0.f, sx = 0.f;
#pragma unroll
    for (int j = 0; j < BN1P_MAXCH; ++j) {
        if (tid + j * BN1P_THREADS < total) {
            float f[8], g[8];
            unpack16<T>(xv[j], f);
            unpack16<T>(dv[j], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s += g[e];
                sx += g[e] * ((f[e] - mu) * is);
            }
        }
    }
    double ds = (double)s, dsx = (double)sx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ds += __shfl_xor(ds, o, 64);
        dsx += __shfl_xor(dsx, o, 64);
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = ds;
        red[1][tid >> 6] = dsx;
    }
    __syncthreads();
    double ts = 0.0, tsx = 0.0;
#pragma unroll
    for (int w = 0; w < BN1P_THREADS / 64; ++w) {
        ts += red[0][w];
        tsx += red[1][w];
    }
    if (tid == 0) {
        if (dgamma) dgamma[c] = (float)tsx;
        if (dbeta) dbeta[c] = (float)ts;
    }
    const float ka = training ? (float)(ts / M) : 0.f, kb = training ? (float)(tsx / M) : 0.f;
#pragma unroll
    for (int j = 0; j < BN1P_MAXCH; ++j) {
        if (tid + j * BN1P_THREADS < total) {
            float f[8], g[8], o[8];
            unpack16<T>(xv[j], f);
            unpack16<T>(dv[j], g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = k1 * (g[e] - ka - (f[e] - mu) * is * kb);
            *reinterpret_cast<uint4*>(dx + off[j]) = pack16<T>(o);
        }
    }
}

// dgamma / dbeta and the BwdXf coefficients of channel c from the reduction partials (fixed order; the Pred loads are
// requested together)
__global__ void __launch_bounds__(64) bn_bwd_coef_kernel(const double* __restrict__ partial, int Pred, int C, double M,
                                                         int training, const float* __restrict__ scale,
                                                         const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, float* __restrict__ ka,
                                                         float* __restrict__ kbi) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, sx = 0.0;
    for (int q0 = 0; q0 < Pred; q0 += 8) {
        double a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + j < Pred ? q0 + j : Pred - 1;
            a[j] = partial[((long long)q * C + c) * 2];
            b[j] = partial[((long long)q * C + c) * 2 + 1];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s += q0 + j < Pred ? a[j] : 0.0;
            sx += q0 + j < Pred ? b[j] : 0.0;
        }
    }
    if (dgamma) dgamma[c] = (float)sx;
    if (dbeta) dbeta[c] = (float)s;
    const double sc = (double)scale[c];
    ka[c] = training ? (float)(sc * (s / M)) : 0.f;
    kbi[c] = training ? (float)(sc * (double)invstd[c] * (sx / M)) : 0.f;
}

// the same from the [C][P] (sum dz, sum dz * (y - mean)) partials a producer kernel left (BwdStatOut): one wave per
// channel, lane l folds partials l, l + 64, .. in fp64, then a fixed butterfly
__global__ void __launch_bounds__(64) bn_bwd_coef_cp_kernel(const float2* __restrict__ partial, int P, int C, double M,
                                                            int training, const float* __restrict__ scale,
                                                            const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ ka,
                                                            float* __restrict__ kbi) {
    const int c = blockIdx.x;
    const float2* pc = partial + (long long)c * P;
    double s = 0.0, st = 0.0;
    for (int q0 = threadIdx.x; q0 < P; q0 += 64 * 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = q0 + 64 * j;
            v[j] = q < P ? pc[q] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s += (double)v[j].x;
            st += (double)v[j].y;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        st += __shfl_xor(st, off, 64);
    }
    if (threadIdx.x == 0) {
        const double is = (double)invstd[c], sc = (double)scale[c];
        const double sx = st * is;          // sum dz * xhat
        if (dgamma) dgamma[c] = (float)sx;
        if (dbeta) dbeta[c] = (float)s;
        ka[c] = training ? (float)(sc * (s / M)) : 0.f;
        kbi[c] = training ? (float)(sc * is * (sx / M)) : 0.f;
    }
}

// ------------------------------------------------------------------- BN backward through PixelShuffle(2)
// ConvLayer with act_func "pixelshuffle" (conv -> BN -> PixelShuffle(2), reference ofa/layers.py:120-151): the incoming
// gradient arrives in the shuffled layout dout[N, C/4, 2H, 2W]; dz[n, 4g + j, h, w] = dout[n, g, 2h + (j >> 1), 2w + (j & 1)].
// Both passes read it THROUGH the inverse shuffle (two 32-byte runs of two up-sampled rows = the same 8 pixels of the four
// channels of group g) instead of a separate un-shuffle pass that writes and re-reads the whole tensor.  16-bit tensors, no
// activation (the shuffle is the layer's activation), W % 8 == 0.  One block = one channel group x images p, p+P, ..
__device__ __forceinline__ uint32_t bnps_lo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
__device__ __forceinline__ uint32_t bnps_hi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }
__device__ __forceinline__ void bnps_unzip(const uint4& lo, const uint4& hi, uint4& a, uint4& b) {
    a = make_uint4(bnps_lo(lo.x, lo.y), bnps_lo(lo.z, lo.w), bnps_lo(hi.x, hi.y), bnps_lo(hi.z, hi.w));
    b = make_uint4(bnps_hi(lo.x, lo.y), bnps_hi(lo.z, lo.w), bnps_hi(hi.x, hi.y), bnps_hi(hi.z, hi.w));
}
// the four channels' 8-pixel pieces of item (h, wq) of group plane `gp` (= n * C/4 + g) from the shuffled tensor
__device__ __forceinline__ void bnps_load(const uint4* __restrict__ dout, long long gp, int h, int wq, int H, int Wq,
                                          uint4 (&dz)[4]) {
    const long long hr0 = gp * 4 * (long long)H * Wq + (long long)(2 * h) * (2 * Wq) + 2 * wq;
    const uint4 lo0 = dout[hr0], hi0 = dout[hr0 + 1], lo1 = dout[hr0 + 2 * Wq], hi1 = dout[hr0 + 2 * Wq + 1];
    bnps_unzip(lo0, hi0, dz[0], dz[1]);
    bnps_unzip(lo1, hi1, dz[2], dz[3]);
}

template <typename T>
__global__ void __launch_bounds__(BN_THREADS) bn_bwd_reduce_ps_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ x,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      double* __restrict__ partial, int N, int C, int H,
                                                                      int Wq, int P) {
    const int g = blockIdx.x, p = blockIdx.y, G = C / 4;
    float mu[4], is[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = mean[4 * g + j];
        is[j] = invstd[4 * g + j];
    }
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
    const int items = H * Wq;
    const long long plane = (long long)H * Wq;
    for (int n = p; n < N; n += P) {
        for (int it = threadIdx.x; it < items; it += BN_THREADS) {
            const int h = it / Wq, wq = it - h * Wq;
            uint4 dz[4], xv[4];
            bnps_load(dout, (long long)n * G + g, h, wq, H, Wq, dz);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = x[((long long)n * C + 4 * g + j) * plane + it];
            // all 8 requests of the item before the first use: with the x load inside the channel loop the compiler issued
            // x[1..3] one at a time, each behind s_waitcnt vmcnt(0) -- 4 serial round trips per item, 0.98 TB/s
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f[8], gd[8];
                unpack16<T>(xv[j], f);
                unpack16<T>(dz[j], gd);
                float s8 = 0.f, q8 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s8 += gd[e];
                    q8 = fmaf(gd[e], (f[e] - mu[j]) * is[j], q8);
                }
                s[j] += (double)s8;
                sx[j] += (double)q8;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        block_reduce2(s[j], sx[j]);
        if (threadIdx.x == 0) {
            partial[((long long)p * C + 4 * g + j) * 2] = s[j];
            partial[((long long)p * C + 4 * g + j) * 2 + 1] = sx[j];
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(BN_THREADS) bn_bwd_apply_ps_kernel(const uint4* __restrict__ dout, const uint4* __restrict__ x,
                                                                     uint4* __restrict__ dx, const float* __restrict__ scale,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const double* __restrict__ partial, int Pred, double M,
                                                                     int training, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int N, int C, int H, int Wq,
                                                                     int P) {
    const int g = blockIdx.x, p = blockIdx.y, G = C / 4;
    float k1[4], mu[4], is[4], ka[4], kb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * g + j;
        double s = 0.0, sx = 0.0;
        for (int q = 0; q < Pred; ++q) {      // Pred <= 16: fixed order, as bn_bwd_apply_kernel
            s += partial[((long long)q * C + c) * 2];
            sx += partial[((long long)q * C + c) * 2 + 1];
        }
        if (p == 0 && threadIdx.x == 0) {
            if (dgamma) dgamma[c] = (float)sx;
            if (dbeta) dbeta[c] = (float)s;
        }
        k1[j] = scale[c];
        mu[j] = mean[c];
        is[j] = invstd[c];
        ka[j] = training ? (float)(s / M) : 0.f;
        kb[j] = training ? (float)(sx / M) : 0.f;
    }
    const int items = H * Wq;
    const long long plane = (long long)H * Wq;
    for (int n = p; n < N; n += P) {
        for (int it = threadIdx.x; it < items; it += BN_THREADS) {
            const int h = it / Wq, wq = it - h * Wq;
            uint4 dz[4], xv[4];
            bnps_load(dout, (long long)n * G + g, h, wq, H, Wq, dz);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = x[((long long)n * C + 4 * g + j) * plane + it];
            __builtin_amdgcn_sched_barrier(0);   // all 8 requests of the item before the first use (as the reduction pass)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long off = ((long long)n * C + 4 * g + j) * plane + it;
                float f[8], gd[8], o[8];
                unpack16<T>(xv[j], f);
                unpack16<T>(dz[j], gd);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = k1[j] * (gd[e] - ka[j] - (f[e] - mu[j]) * is[j] * kb[j]);
                dx[off] = pack16<T>(o);
            }
        }
    }
}

static int bn_parts(int64_t N, int64_t C) {
    // enough blocks to fill the chip (256 CUs x ~8 blocks), at most one image per part
    int64_t want = cdiv(2048, C > 0 ? C : 1);
    if (want < 1) want = 1;
    if (want > N) want = N;
    return (int)want;
}

static int check_bn(const char* name, int64_t N, int64_t C, int64_t HW, int dtype) {
    OFASR_REQUIRE(N > 0 && C > 0 && HW > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape N=%lld C=%lld HW=%lld", name,
                  (long long)N, (long long)C, (long long)HW);
    OFASR_REQUIRE(dtype == OFASR_F32 || dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_INVALID_ARG,
                  "%s: bad dtype %d", name, dtype);
    OFASR_REQUIRE(N <= INT32_MAX && C <= 65535 * 32 && HW <= INT32_MAX, OFASR_ERR_UNSUPPORTED, "%s: too large", name);
    return OFASR_OK;
}

static bool vec_ok(int64_t HW, int dtype, const void* a, const void* b, const void* c, const void* d) {
    const int V = dtype == OFASR_F32 ? 4 : 8;
    uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                     reinterpret_cast<uintptr_t>(d);
    return (HW % V) == 0 && (bits & 15) == 0;
}

}  // namespace ofasr

using namespace ofasr;

#define OFASR_BN_DISPATCH_T(dtype, CALL)      \
    switch (dtype) {                          \
        case OFASR_F32: { using T = float; CALL; } break;   \
        case OFASR_F16: { using T = f16_t; CALL; } break;   \
        default: { using T = bf16_t; CALL; } break;         \
    }

OFASR_EXPORT size_t ofasr_bn_workspace(int64_t N, int64_t C) {
    if (N <= 0 || C <= 0) return 0;
    return (size_t)bn_parts(N, C) * (size_t)C * 2 * sizeof(double);
}

OFASR_EXPORT int ofasr_bn_stats(const void* x, int64_t N, int64_t C, int64_t HW, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_bn_stats";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(x && workspace && workspace_bytes >= ofasr_bn_workspace(N, C), OFASR_ERR_WORKSPACE,
                  "%s: null input or workspace too small", name);
    const int P = bn_parts(N, C);
    dim3 grid((unsigned)C, (unsigned)P);
    hipStream_t st = as_stream(stream);
    const bool v = vec_ok(HW, dtype, x, nullptr, nullptr, nullptr);
    prof_note((dtype == OFASR_F32 ? 4.0 : 2.0) * (double)N * (double)C * (double)HW, 0.0);
    OFASR_BN_DISPATCH_T(dtype, {
        if (v) OFASR_LAUNCH((bn_stats_kernel<T, true>), grid, dim3(BN_THREADS), 0, st, (const T*)x,
                                  (double*)workspace, (int)N, (int)C, (int)HW, P);
        else OFASR_LAUNCH((bn_stats_kernel<T, false>), grid, dim3(BN_THREADS), 0, st, (const T*)x,
                                (double*)workspace, (int)N, (int)C, (int)HW, P);
    });
    return check_launch(name);
}

OFASR_EXPORT int ofasr_bn_finalize(const void* workspace, int64_t n_partials, int64_t C, double count,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   double momentum, double eps, int training, float* mean, float* invstd, float* scale,
                                   float* shift, void* stream) {
    const char* name = "ofasr_bn_finalize";
    OFASR_REQUIRE(C > 0 && mean && invstd && scale && shift, OFASR_ERR_INVALID_ARG, "%s: null output or C<=0", name);
    OFASR_REQUIRE(training ? (workspace != nullptr && n_partials > 0 && count > 0) : (running_mean && running_var),
                  OFASR_ERR_INVALID_ARG, "%s: missing statistics source", name);
    OFASR_LAUNCH(bn_finalize_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream),
                       (const double*)workspace, (int)n_partials, (int)C, count, gamma, beta, running_mean, running_var,
                       momentum, eps, training, mean, invstd, scale, shift);
    return check_launch(name);
}

static int launch_bn_apply(const char* name, const void* x, const void* residual, void* y, const BnSource& src,
                           int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream);
namespace ofasr {
// BatchNorm (+act)(+residual) forward whose statistics come from conv-epilogue partials: the finalize is folded into
// every block of the apply kernel (no finalize launch).  training == 0: running statistics.
int bn_fwd_cp(const void* x, const void* residual, void* y, const float2* partial, int64_t P, const float* gamma,
              const float* beta, float* running_mean, float* running_var, double momentum, double eps, int training,
              float* stats, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream) {
    const char* name = "bn_fwd_cp";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(x && y && stats, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(training ? (partial != nullptr && P > 0) : (running_mean && running_var), OFASR_ERR_INVALID_ARG,
                  "%s: missing statistics source", name);
    BnSource src{};
    src.mode = training ? 3 : 2;
    src.cp = partial;
    src.Pstat = (int)P;
    src.M = (double)N * (double)HW;
    src.momentum = momentum;
    src.eps = eps;
    src.gamma = gamma;
    src.beta = beta;
    src.running_mean = running_mean;
    src.running_var = running_var;
    src.mean = stats;
    src.invstd = stats + C;
    src.scale = stats + 2 * C;
    src.shift = stats + 3 * C;
    return launch_bn_apply(name, x, residual, y, src, N, C, HW, act, dtype, stream);
}

// one wave per channel: lane l sums partials l, l+64, ... in fp64, then a fixed-order butterfly
__global__ void __launch_bounds__(64) bn_finalize_cp_kernel(const float2* __restrict__ partial, int P, int C, double M,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, double momentum, double eps,
                                                            int training, float* __restrict__ mean_out,
                                                            float* __restrict__ invstd_out, float* __restrict__ scale,
                                                            float* __restrict__ shift, int64_t* k0, int64_t* k1,
                                                            int64_t* k2) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0) {
        if (k0) *k0 += 1;
        if (k1) *k1 += 1;
        if (k2) *k2 += 1;
    }
    double mean, var;
    if (training) {
        double s = 0.0, ss = 0.0;
        for (int p = lane; p < P; p += 64) {
            const float2 v = partial[(long long)c * P + p];
            s += (double)v.x;
            ss += (double)v.y;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            ss += __shfl_xor(ss, o, 64);
        }
        mean = s / M;
        var = ss / M - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    if (lane != 0) return;
    if (training && running_mean) {
        const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
    const double invstd = 1.0 / sqrt(var + eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    mean_out[c] = (float)mean;
    invstd_out[c] = (float)invstd;
    scale[c] = (float)(g * invstd);
    shift[c] = (float)(b - mean * g * invstd);
}

int bn_finalize_cp(const float2* partial, int64_t P, int64_t C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, double momentum, double eps, int training, float* mean,
                   float* invstd, float* scale, float* shift, int64_t* k0, int64_t* k1, int64_t* k2, void* stream) {
    const char* name = "bn_finalize_cp";
    OFASR_REQUIRE(C > 0 && mean && invstd && scale && shift, OFASR_ERR_INVALID_ARG, "%s: null output or C<=0", name);
    OFASR_REQUIRE(training ? (partial != nullptr && P > 0 && count > 0) : (running_mean && running_var),
                  OFASR_ERR_INVALID_ARG, "%s: missing statistics source", name);
    OFASR_LAUNCH(bn_finalize_cp_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(stream), partial, (int)P, (int)C,
                       count, gamma, beta, running_mean, running_var, momentum, eps, training, mean, invstd, scale, shift,
                       k0, k1, k2);
    return check_launch(name);
}

int bn_finalize_bump(const void* workspace, int64_t n_partials, int64_t C, double count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                     int training, float* mean, float* invstd, float* scale, float* shift, int64_t* k0, int64_t* k1,
                     int64_t* k2, void* stream) {
    const char* name = "bn_finalize_bump";
    OFASR_REQUIRE(C > 0 && mean && invstd && scale && shift, OFASR_ERR_INVALID_ARG, "%s: null output or C<=0", name);
    OFASR_REQUIRE(training ? (workspace != nullptr && n_partials > 0 && count > 0) : (running_mean && running_var),
                  OFASR_ERR_INVALID_ARG, "%s: missing statistics source", name);
    OFASR_LAUNCH(bn_finalize_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, as_stream(stream),
                       (const double*)workspace, (int)n_partials, (int)C, count, gamma, beta, running_mean, running_var,
                       momentum, eps, training, mean, invstd, scale, shift, k0, k1, k2);
    return check_launch(name);
}
}  // namespace ofasr

OFASR_EXPORT int ofasr_bn_partials(int64_t N, int64_t C) { return (N > 0 && C > 0) ? bn_parts(N, C) : 0; }

static int launch_bn_apply(const char* name, const void* x, const void* residual, void* y, const BnSource& src,
                           int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream) {
    const int P = bn_parts(N, C);
    dim3 grid((unsigned)C, (unsigned)P);
    hipStream_t st = as_stream(stream);
    const bool v = vec_ok(HW, dtype, x, residual, y, nullptr);
    prof_note((dtype == OFASR_F32 ? 4.0 : 2.0) * (double)N * (double)C * (double)HW * (residual ? 3.0 : 2.0), 0.0);
#define OFASR_BNF(VEC, ACT, RES)                                                                                   \
    OFASR_LAUNCH((bn_act_fwd_kernel<T, VEC, ACT, RES>), grid, dim3(BN_THREADS), 0, st, (const T*)x,          \
                       (const T*)residual, (T*)y, src, (int)N, (int)C, (int)HW, P)
    OFASR_BN_DISPATCH_T(dtype, {
        if (v) {
            if (act == 1) { if (residual) OFASR_BNF(true, 1, true); else OFASR_BNF(true, 1, false); }
            else { if (residual) OFASR_BNF(true, 0, true); else OFASR_BNF(true, 0, false); }
        } else {
            if (act == 1) { if (residual) OFASR_BNF(false, 1, true); else OFASR_BNF(false, 1, false); }
            else { if (residual) OFASR_BNF(false, 0, true); else OFASR_BNF(false, 0, false); }
        }
    });
#undef OFASR_BNF
    return check_launch(name);
}

OFASR_EXPORT int ofasr_bn_act_fwd(const void* x, const void* residual, void* y, const float* scale, const float* shift,
                                  const float* mean, int64_t N, int64_t C, int64_t HW, int act, int dtype,
                                  void* stream) {
    const char* name = "ofasr_bn_act_fwd";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(x && y && scale && shift && mean, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(act == 0 || act == 1, OFASR_ERR_UNSUPPORTED, "%s: act %d not in {0 none, 1 relu6}", name, act);
    BnSource src{};
    src.mode = 0;
    src.scale = const_cast<float*>(scale);
    src.shift = const_cast<float*>(shift);
    src.mean = const_cast<float*>(mean);
    return launch_bn_apply(name, x, residual, y, src, N, C, HW, act, dtype, stream);
}

// statistics pass (training) + apply pass with the finalize folded in: 2 launches (1 in eval mode)
OFASR_EXPORT int ofasr_bn_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, double momentum, double eps, int training,
                              float* stats, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* workspace,
                              size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_bn_fwd";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(x && y && stats, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(act == 0 || act == 1, OFASR_ERR_UNSUPPORTED, "%s: act %d not in {0 none, 1 relu6}", name, act);
    OFASR_REQUIRE(training || (running_mean && running_var), OFASR_ERR_INVALID_ARG, "%s: eval mode needs running stats",
                  name);
    BnSource src{};
    src.mode = training ? 1 : 2;
    src.M = (double)N * (double)HW;
    src.momentum = momentum;
    src.eps = eps;
    src.gamma = gamma;
    src.beta = beta;
    src.running_mean = running_mean;
    src.running_var = running_var;
    src.mean = stats;
    src.invstd = stats + C;
    src.scale = stats + 2 * C;
    src.shift = stats + 3 * C;
    if (training) {
        rc = ofasr_bn_stats(x, N, C, HW, dtype, workspace, workspace_bytes, stream);
        if (rc) return rc;
        src.partial = (const double*)workspace;
        src.Pstat = bn_parts(N, C);
    }
    return launch_bn_apply(name, x, residual, y, src, N, C, HW, act, dtype, stream);
}

// BatchNorm forward whose batch statistics come from the producing conv's epilogue partials ([C][P] (sum, sum of squares),
// e.g. ofasr_conv2d_fwd_stat): no pass over x for statistics.
//   ofasr_bn_fwd_cp       apply (+act, +residual) with the fold of the partials in every block of the apply kernel; stats
//                         [4*C] receives mean | invstd | scale | shift, running statistics are updated (training)
//   ofasr_bn_finalize_cp  only the fold: stats + running statistics (the apply is done by another kernel, e.g.
//                         ofasr_pixel_shuffle2_bn)
OFASR_EXPORT int ofasr_bn_fwd_cp(const void* x, const void* residual, void* y, const void* partial, int64_t P,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 double momentum, double eps, int training, float* stats, int64_t N, int64_t C, int64_t HW,
                                 int act, int dtype, void* stream) {
    OFASR_REQUIRE(act == 0 || act == 1, OFASR_ERR_UNSUPPORTED, "ofasr_bn_fwd_cp: act %d not in {0 none, 1 relu6}", act);
    return bn_fwd_cp(x, residual, y, (const float2*)partial, P, gamma, beta, running_mean, running_var, momentum, eps, training,
                     stats, N, C, HW, act, dtype, stream);
}

OFASR_EXPORT int ofasr_bn_finalize_cp(const void* partial, int64_t P, int64_t C, double count, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                                      int training, float* stats, void* stream) {
    OFASR_REQUIRE(stats != nullptr && C > 0, OFASR_ERR_INVALID_ARG, "ofasr_bn_finalize_cp: null stats");
    return bn_finalize_cp((const float2*)partial, P, C, count, gamma, beta, running_mean, running_var, momentum, eps, training,
                          stats, stats + C, stats + 2 * C, stats + 3 * C, nullptr, nullptr, nullptr, stream);
}

OFASR_EXPORT int ofasr_bn_act_bwd(const void* dy, const void* x, const void* residual, void* dx, void* dresidual,
                                  const float* scale, const float* shift, const float* mean, const float* invstd,
                                  float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW, int act, int training,
                                  int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_bn_act_bwd";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(dy && x && dx && scale && shift && mean && invstd, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(act == 0 || act == 1, OFASR_ERR_UNSUPPORTED, "%s: act %d not in {0 none, 1 relu6}", name, act);
    const int P = bn_parts(N, C);
    const size_t need = (size_t)P * C * 2 * sizeof(double);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    double* partial = (double*)workspace;
    const double M = (double)N * (double)HW;
    dim3 grid((unsigned)C, (unsigned)P);
    hipStream_t st = as_stream(stream);
    const bool v = vec_ok(HW, dtype, dy, x, residual, dx) && ((reinterpret_cast<uintptr_t>(dresidual) & 15) == 0);
    {
        // small channels: reduction and apply by one workgroup per channel from registers (bn_bwd_onepass_kernel)
        // OFF by default (OFASR_BN_BWD_ONEPASS=1): in the training step 2604-2614 against 2686-2695 images/s (two A/B pairs)
        // -- 64 workgroups of 1024 threads are a latency-bound kernel on a quarter of the CUs and cannot share them with the
        // side stream's kernels; the two 7 us passes on the whole chip are shorter
        static const bool onepass = [] { const char* e = getenv("OFASR_BN_BWD_ONEPASS"); return e && e[0] == '1'; }();
        if (onepass && v && act == 0 && !residual && !dresidual && dtype != OFASR_F32 && HW % 8 == 0 &&
            N * (HW / 8) <= (int64_t)BN1P_THREADS * BN1P_MAXCH && C <= 65535) {
            prof_note(2.0 * (double)N * (double)C * (double)HW * 3.0, 0.0);       // reads dy, x; writes dx
            if (dtype == OFASR_BF16)
                OFASR_LAUNCH((bn_bwd_onepass_kernel<bf16_t>), dim3((unsigned)C), dim3(BN1P_THREADS), 0, st, (const bf16_t*)dy,
                             (const bf16_t*)x, (bf16_t*)dx, scale, mean, invstd, dgamma, dbeta, (int)N, (int)C, (int)HW, M,
                             training);
            else
                OFASR_LAUNCH((bn_bwd_onepass_kernel<f16_t>), dim3((unsigned)C), dim3(BN1P_THREADS), 0, st, (const f16_t*)dy,
                             (const f16_t*)x, (f16_t*)dx, scale, mean, invstd, dgamma, dbeta, (int)N, (int)C, (int)HW, M,
                             training);
            return check_launch(name);
        }
    }
#define OFASR_BNR(VEC, ACT, RES)                                                                                     \
    OFASR_LAUNCH((bn_bwd_reduce_kernel<T, VEC, ACT, RES>), grid, dim3(BN_THREADS), 0, st, (const T*)dy,        \
                       (const T*)x, (const T*)residual, scale, shift, mean, invstd, partial, (int)N, (int)C, (int)HW, P)
#define OFASR_BNA(VEC, ACT, RES)                                                                                     \
    OFASR_LAUNCH((bn_bwd_apply_kernel<T, VEC, ACT, RES>), grid, dim3(BN_THREADS), 0, st, (const T*)dy,         \
                       (const T*)x, (const T*)residual, (T*)dx, (T*)dresidual, scale, shift, mean, invstd,           \
                       (const double*)partial, P, M, training, dgamma, dbeta, (int)N, (int)C, (int)HW, P)
#define OFASR_BN_BOTH(MACRO)                                                                              \
    OFASR_BN_DISPATCH_T(dtype, {                                                                          \
        if (v) {                                                                                          \
            if (act == 1) { if (residual) MACRO(true, 1, true); else MACRO(true, 1, false); }             \
            else { if (residual) MACRO(true, 0, true); else MACRO(true, 0, false); }                      \
        } else {                                                                                          \
            if (act == 1) { if (residual) MACRO(false, 1, true); else MACRO(false, 1, false); }           \
            else { if (residual) MACRO(false, 0, true); else MACRO(false, 0, false); }                    \
        }                                                                                                 \
    })
    const double tensor_bytes = (dtype == OFASR_F32 ? 4.0 : 2.0) * (double)N * (double)C * (double)HW;
    prof_note(tensor_bytes * (residual ? 3.0 : 2.0), 0.0);                         // reads dy, x (, residual)
    OFASR_BN_BOTH(OFASR_BNR);
    rc = check_launch(name);
    if (rc) return rc;
    prof_note(tensor_bytes * (3.0 + (residual ? 1.0 : 0.0) + (dresidual ? 1.0 : 0.0)), 0.0);   // + writes dx (, dresidual)
    OFASR_BN_BOTH(OFASR_BNA);
#undef OFASR_BNR
#undef OFASR_BNA
#undef OFASR_BN_BOTH
    return check_launch(name);
}

namespace ofasr {
int bn_bwd_coef_cp(const float2* partial, int64_t P, int64_t C, double count, int training, const float* scale,
                   const float* invstd, float* dgamma, float* dbeta, float* ka, float* kbi, void* stream) {
    const char* name = "bn_bwd_coef_cp";
    OFASR_REQUIRE(partial && scale && invstd && ka && kbi, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(P > 0 && P <= INT32_MAX && C > 0 && C <= INT32_MAX && count > 0, OFASR_ERR_INVALID_ARG, "%s: bad shape", name);
    OFASR_LAUNCH(bn_bwd_coef_cp_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(stream), partial, (int)P, (int)C, count,
                 training, scale, invstd, dgamma, dbeta, ka, kbi);
    return check_launch(name);
}

int bn_bwd_reduce_coef(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* invstd, float* dgamma, float* dbeta, float* ka, float* kbi, int64_t N, int64_t C,
                       int64_t HW, int act, int training, int dtype, void* workspace, size_t workspace_bytes,
                       void* stream) {
    const char* name = "bn_bwd_reduce_coef";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(dy && x && scale && shift && mean && invstd && ka && kbi, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    const int P = bn_parts(N, C);
    const size_t need = (size_t)P * C * 2 * sizeof(double);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    double* partial = (double*)workspace;
    dim3 grid((unsigned)C, (unsigned)P);
    hipStream_t st = as_stream(stream);
    const bool v = vec_ok(HW, dtype, dy, x, nullptr, nullptr);
    const void* residual = nullptr;
    prof_note((dtype == OFASR_F32 ? 4.0 : 2.0) * (double)N * (double)C * (double)HW * 2.0, 0.0);
#define OFASR_BNR2(VEC, ACT)                                                                                           \
    OFASR_LAUNCH((bn_bwd_reduce_kernel<T, VEC, ACT, false>), grid, dim3(BN_THREADS), 0, st, (const T*)dy, (const T*)x, \
                 (const T*)residual, scale, shift, mean, invstd, partial, (int)N, (int)C, (int)HW, P)
    OFASR_BN_DISPATCH_T(dtype, {
        if (v) { if (act == 1) OFASR_BNR2(true, 1); else OFASR_BNR2(true, 0); }
        else { if (act == 1) OFASR_BNR2(false, 1); else OFASR_BNR2(false, 0); }
    });
#undef OFASR_BNR2
    rc = check_launch(name);
    if (rc) return rc;
    OFASR_LAUNCH(bn_bwd_coef_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, st, (const double*)partial, P, (int)C,
                 (double)N * (double)HW, training, scale, invstd, dgamma, dbeta, ka, kbi);
    return check_launch(name);
}
int bn_bwd_reduce_only(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                       const float* invstd, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* workspace,
                       size_t workspace_bytes, int* P_out, void* stream) {
    const char* name = "bn_bwd_reduce_only";
    int rc = check_bn(name, N, C, HW, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(dy && x && scale && shift && mean && invstd && P_out, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    const int P = bn_parts(N, C);
    const size_t need = (size_t)P * C * 2 * sizeof(double);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    double* partial = (double*)workspace;
    dim3 grid((unsigned)C, (unsigned)P);
    hipStream_t st = as_stream(stream);
    const bool v = vec_ok(HW, dtype, dy, x, nullptr, nullptr);
    const void* residual = nullptr;
    prof_note((dtype == OFASR_F32 ? 4.0 : 2.0) * (double)N * (double)C * (double)HW * 2.0, 0.0);
#define OFASR_BNR2(VEC, ACT)                                                                                           \
    OFASR_LAUNCH((bn_bwd_reduce_kernel<T, VEC, ACT, false>), grid, dim3(BN_THREADS), 0, st, (const T*)dy, (const T*)x, \
                 (const T*)residual, scale, shift, mean, invstd, partial, (int)N, (int)C, (int)HW, P)
    OFASR_BN_DISPATCH_T(dtype, {
        if (v) { if (act == 1) OFASR_BNR2(true, 1); else OFASR_BNR2(true, 0); }
        else { if (act == 1) OFASR_BNR2(false, 1); else OFASR_BNR2(false, 0); }
    });
#undef OFASR_BNR2
    *P_out = P;
    return check_launch(name);
}
}  // namespace ofasr

OFASR_EXPORT size_t ofasr_bn_bwd_ps2_workspace(int64_t N, int64_t C) {
    if (N <= 0 || C < 4) return 0;
    return (size_t)bn_parts(N, C / 4) * C * 2 * sizeof(double);
}

// BatchNorm backward of a layer whose output went through PixelShuffle(2): dout is the gradient in the SHUFFLED layout
// [N, C/4, 2H, 2W]; dx, dgamma, dbeta as ofasr_bn_act_bwd with act = 0 and no residual.  f16 / bf16, W % 8 == 0,
// C % 4 == 0, 16-byte aligned tensors; workspace: ofasr_bn_bwd_ps2_workspace(N, C) bytes.
OFASR_EXPORT int ofasr_bn_bwd_ps2(const void* dout, const void* x, void* dx, const float* scale, const float* mean,
                                  const float* invstd, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t H, int64_t W,
                                  int training, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_bn_bwd_ps2";
    int rc = check_bn(name, N, C, H * W, dtype);
    if (rc) return rc;
    OFASR_REQUIRE(dout && x && dx && scale && mean && invstd, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    OFASR_REQUIRE(dtype == OFASR_F16 || dtype == OFASR_BF16, OFASR_ERR_UNSUPPORTED, "%s: 16-bit tensors only", name);
    OFASR_REQUIRE(C % 4 == 0 && W % 8 == 0 && H <= (1 << 20) && W <= (1 << 20), OFASR_ERR_UNSUPPORTED,
                  "%s: needs C %% 4 == 0 and W %% 8 == 0", name);
    OFASR_REQUIRE(((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
                  OFASR_ERR_UNSUPPORTED, "%s: tensors must be 16-byte aligned", name);
    const int P = bn_parts(N, C / 4);
    const size_t need = (size_t)P * C * 2 * sizeof(double);
    OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B", name,
                  workspace_bytes, need);
    double* partial = (double*)workspace;
    const int Wq = (int)(W / 8);
    const double M = (double)N * (double)H * (double)W;
    dim3 grid((unsigned)(C / 4), (unsigned)P);
    hipStream_t st = as_stream(stream);
    const double tensor_bytes = 2.0 * (double)N * (double)C * (double)H * (double)W;
    prof_note(2.0 * tensor_bytes, 0.0);
    if (dtype == OFASR_BF16)
        OFASR_LAUNCH((bn_bwd_reduce_ps_kernel<bf16_t>), grid, dim3(BN_THREADS), 0, st, (const uint4*)dout, (const uint4*)x, mean,
                     invstd, partial, (int)N, (int)C, (int)H, Wq, P);
    else
        OFASR_LAUNCH((bn_bwd_reduce_ps_kernel<f16_t>), grid, dim3(BN_THREADS), 0, st, (const uint4*)dout, (const uint4*)x, mean,
                     invstd, partial, (int)N, (int)C, (int)H, Wq, P);
    rc = check_launch(name);
    if (rc) return rc;
    prof_note(3.0 * tensor_bytes, 0.0);
    if (dtype == OFASR_BF16)
        OFASR_LAUNCH((bn_bwd_apply_ps_kernel<bf16_t>), grid, dim3(BN_THREADS), 0, st, (const uint4*)dout, (const uint4*)x,
                     (uint4*)dx, scale, mean, invstd, (const double*)partial, P, M, training, dgamma, dbeta, (int)N, (int)C,
                     (int)H, Wq, P);
    else
        OFASR_LAUNCH((bn_bwd_apply_ps_kernel<f16_t>), grid, dim3(BN_THREADS), 0, st, (const uint4*)dout, (const uint4*)x,
                     (uint4*)dx, scale, mean, invstd, (const double*)partial, P, M, training, dgamma, dbeta, (int)N, (int)C,
                     (int)H, Wq, P);
    return check_launch(name);
}

OFASR_EXPORT size_t ofasr_bn_act_bwd_workspace(int64_t N, int64_t C) {
    if (N <= 0 || C <= 0) return 0;
    return (size_t)bn_parts(N, C) * C * 2 * sizeof(double);
}
