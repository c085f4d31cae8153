/*
 * ofasr_oracle.c -- CPU restatement of the OFA-SR hot-path operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The shipped path is the HIP library (ofa-for-super-resolution_amd/csrc).
 *
 * Every function restates, as plain loops, what the reference gets from an ATen call.  The
 * reference itself holds no tests or golden vectors, so this oracle is
 * pinned by fixtures generated from the reference imported in the build container
 * (tests/golden/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).
 *
 * Arithmetic: inputs/outputs are fp32 NCHW-contiguous like the reference's tensors; sums are
 * carried in double and rounded once, so the oracle sits within 1 ulp of the exact result and
 * both oneDNN (reference, CPU) and the HIP kernels are compared against it with a
 * reassociation tolerance.  PixelShuffle/Unshuffle move bytes and are bit-exact.
 *
 * Reference call sites restated (paths relative to /root/reference):
 *   pwconv_*      ofa/elastic_nn/modules/dynamic_op.py:104-112  (weight[:out,:in] slice + F.conv2d 1x1)
 *   dwconv_*      ofa/elastic_nn/modules/dynamic_op.py:73-84    (F.conv2d groups=C, pad=k//2, stride 1)
 *   ktransform_*  ofa/elastic_nn/modules/dynamic_op.py:46-71    (centre crop + F.linear chain)
 *                 ofa/imagenet_codebase/utils/__init__.py:89-94 (sub_filter_start_end)
 *   pixel_shuffle ofa/utils.py:259-260,309-310                  (nn.PixelShuffle(2))
 *   pixel_unshuffle ofa/utils.py:383-397                        (one-hot grouped strided conv)
 *   conv2d_*      ofa/layers.py:120-151                         (static ConvLayer nn.Conv2d, same padding)
 *   bn_*          ofa/elastic_nn/modules/dynamic_op.py:148-167  (F.batch_norm on [:dim] slices)
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * PixelShuffle / PixelUnshuffle (pure index permutation, any element size).
 *   shuffle:   y[n, c, h*r+i, w*r+j] = x[n, c*r*r + i*r + j, h, w]      (ofa/utils.py:309-310)
 *   unshuffle: the inverse map; equals the reference's one-hot conv       (ofa/utils.py:383-397)
 * x: [N, C*r*r, H, W]   y: [N, C, H*r, W*r]
 * ---------------------------------------------------------------------------------------- */
ORA_API void ora_pixel_shuffle_fwd(const void* x, void* y, long N, long C, long H, long W,
                                   int r, int esize) {
    const char* xs = (const char*)x;
    char* ys = (char*)y;
    const long Ho = H * r, Wo = W * r;
    for (long n = 0; n < N; ++n)
        for (long c = 0; c < C; ++c)
            for (int i = 0; i < r; ++i)
                for (int j = 0; j < r; ++j)
                    for (long h = 0; h < H; ++h)
                        for (long w = 0; w < W; ++w) {
                            long src = ((n * C * r * r + c * r * r + i * r + j) * H + h) * W + w;
                            long dst = ((n * C + c) * Ho + h * r + i) * Wo + w * r + j;
                            memcpy(ys + dst * esize, xs + src * esize, (size_t)esize);
                        }
}

/* x: [N, C, H*r, W*r] -> y: [N, C*r*r, H, W]  (also the gradient map of ora_pixel_shuffle_fwd) */
ORA_API void ora_pixel_unshuffle_fwd(const void* x, void* y, long N, long C, long H, long W,
                                     int r, int esize) {
    const char* xs = (const char*)x;
    char* ys = (char*)y;
    const long Ho = H * r, Wo = W * r;
    for (long n = 0; n < N; ++n)
        for (long c = 0; c < C; ++c)
            for (int i = 0; i < r; ++i)
                for (int j = 0; j < r; ++j)
                    for (long h = 0; h < H; ++h)
                        for (long w = 0; w < W; ++w) {
                            long dst = ((n * C * r * r + c * r * r + i * r + j) * H + h) * W + w;
                            long src = ((n * C + c) * Ho + h * r + i) * Wo + w * r + j;
                            memcpy(ys + dst * esize, xs + src * esize, (size_t)esize);
                        }
}

/* ------------------------------------------------------------------------------------------
 * Pointwise (1x1) convolution on a weight slice read in place.
 *   y[n,co,p] = sum_ci w[co*ldw + ci] * x[n,ci,p]       co < Cout, ci < Cin
 * w is the max-size parameter [Cout_max, Cin_max(=ldw), 1, 1]; the active slice is its
 * top-left [Cout, Cin] block (dynamic_op.py:108).
 * ---------------------------------------------------------------------------------------- */
ORA_API void ora_pwconv_fwd(const float* x, const float* w, long ldw, float* y,
                            long N, long Cin, long Cout, long HW) {
    double* acc = (double*)malloc(sizeof(double) * (size_t)HW);
    for (long n = 0; n < N; ++n)
        for (long co = 0; co < Cout; ++co) {
            for (long p = 0; p < HW; ++p) acc[p] = 0.0;
            for (long ci = 0; ci < Cin; ++ci) {
                const double wv = (double)w[co * ldw + ci];
                const float* xr = x + (n * Cin + ci) * HW;
                for (long p = 0; p < HW; ++p) acc[p] += wv * (double)xr[p];
            }
            float* yr = y + (n * Cout + co) * HW;
            for (long p = 0; p < HW; ++p) yr[p] = (float)acc[p];
        }
    free(acc);
}

/* dx[n,ci,p] = sum_co w[co*ldw+ci] * dy[n,co,p] */
ORA_API void ora_pwconv_dgrad(const float* dy, const float* w, long ldw, float* dx,
                              long N, long Cin, long Cout, long HW) {
    double* acc = (double*)malloc(sizeof(double) * (size_t)HW);
    for (long n = 0; n < N; ++n)
        for (long ci = 0; ci < Cin; ++ci) {
            for (long p = 0; p < HW; ++p) acc[p] = 0.0;
            for (long co = 0; co < Cout; ++co) {
                const double wv = (double)w[co * ldw + ci];
                const float* dr = dy + (n * Cout + co) * HW;
                for (long p = 0; p < HW; ++p) acc[p] += wv * (double)dr[p];
            }
            float* xr = dx + (n * Cin + ci) * HW;
            for (long p = 0; p < HW; ++p) xr[p] = (float)acc[p];
        }
    free(acc);
}

/* dw[co*ldw+ci] = sum_{n,p} dy[n,co,p] * x[n,ci,p]   (only the [Cout,Cin] slice is written;
 * autograd of the slice leaves exact zeros elsewhere -- the caller pre-zeroes dw). */
ORA_API void ora_pwconv_wgrad(const float* dy, const float* x, float* dw, long ldw,
                              long N, long Cin, long Cout, long HW) {
    for (long co = 0; co < Cout; ++co)
        for (long ci = 0; ci < Cin; ++ci) {
            double s = 0.0;
            for (long n = 0; n < N; ++n) {
                const float* dr = dy + (n * Cout + co) * HW;
                const float* xr = x + (n * Cin + ci) * HW;
                for (long p = 0; p < HW; ++p) s += (double)dr[p] * (double)xr[p];
            }
            dw[co * ldw + ci] = (float)s;
        }
}

/* ------------------------------------------------------------------------------------------
 * Depthwise KxK convolution, stride 1, dilation 1, zero padding K/2 (dynamic_op.py:79-83).
 * f: [C, K, K] (the active filter produced by ora_ktransform_fwd).
 * ---------------------------------------------------------------------------------------- */
ORA_API void ora_dwconv_fwd(const float* x, const float* f, float* y,
                            long N, long C, long H, long W, int K) {
    const int pad = K / 2;
    for (long n = 0; n < N; ++n)
        for (long c = 0; c < C; ++c) {
            const float* xp = x + (n * C + c) * H * W;
            const float* fp = f + c * K * K;
            float* yp = y + (n * C + c) * H * W;
            for (long h = 0; h < H; ++h)
                for (long w = 0; w < W; ++w) {
                    double s = 0.0;
                    for (int i = 0; i < K; ++i) {
                        long hh = h + i - pad;
                        if (hh < 0 || hh >= H) continue;
                        for (int j = 0; j < K; ++j) {
                            long ww = w + j - pad;
                            if (ww < 0 || ww >= W) continue;
                            s += (double)fp[i * K + j] * (double)xp[hh * W + ww];
                        }
                    }
                    yp[h * W + w] = (float)s;
                }
        }
}

/* dx[n,c,h,w] = sum_{i,j} f[c,i,j] * dy[n,c,h-i+pad,w-j+pad] */
ORA_API void ora_dwconv_dgrad(const float* dy, const float* f, float* dx,
                              long N, long C, long H, long W, int K) {
    const int pad = K / 2;
    for (long n = 0; n < N; ++n)
        for (long c = 0; c < C; ++c) {
            const float* dp = dy + (n * C + c) * H * W;
            const float* fp = f + c * K * K;
            float* xp = dx + (n * C + c) * H * W;
            for (long h = 0; h < H; ++h)
                for (long w = 0; w < W; ++w) {
                    double s = 0.0;
                    for (int i = 0; i < K; ++i) {
                        long hh = h - i + pad;
                        if (hh < 0 || hh >= H) continue;
                        for (int j = 0; j < K; ++j) {
                            long ww = w - j + pad;
                            if (ww < 0 || ww >= W) continue;
                            s += (double)fp[i * K + j] * (double)dp[hh * W + ww];
                        }
                    }
                    xp[h * W + w] = (float)s;
                }
        }
}

/* df[c,i,j] = sum_{n,h,w} dy[n,c,h,w] * x[n,c,h+i-pad,w+j-pad] */
ORA_API void ora_dwconv_wgrad(const float* dy, const float* x, float* df,
                              long N, long C, long H, long W, int K) {
    const int pad = K / 2;
    for (long c = 0; c < C; ++c)
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                double s = 0.0;
                for (long n = 0; n < N; ++n) {
                    const float* dp = dy + (n * C + c) * H * W;
                    const float* xp = x + (n * C + c) * H * W;
                    for (long h = 0; h < H; ++h) {
                        long hh = h + i - pad;
                        if (hh < 0 || hh >= H) continue;
                        for (long w = 0; w < W; ++w) {
                            long ww = w + j - pad;
                            if (ww < 0 || ww >= W) continue;
                            s += (double)dp[h * W + w] * (double)xp[hh * W + ww];
                        }
                    }
                }
                df[(c * K + i) * K + j] = (float)s;
            }
}

/* ------------------------------------------------------------------------------------------
 * Elastic-kernel filter: centre crop of the max-size depthwise weight, optionally passed
 * through the learned transform chain (dynamic_op.py:46-71).
 *
 *   ks[0] > ks[1] > ... > ks[nsteps]   the chain of kernel sizes walked, ks[0] = max kernel,
 *                                      ks[nsteps] = active kernel (nsteps = 0: plain w_max).
 *   mats[s]                            the '%dto%d_matrix' for step s: [ks[s+1]^2, ks[s+1]^2],
 *                                      applied as F.linear: out[c,t] = sum_u in[c,u] * M[t,u]
 *                                      where in = centre crop (ks[s+1]) of the current filter.
 *   transform == 0                     KERNEL_TRANSFORM_MODE is None: f = centre crop of w_max
 *                                      to ks[nsteps] (mats ignored).
 * w_max: [Cmax, kmax, kmax] (rows c < C used)   f: [C, K, K]
 * ---------------------------------------------------------------------------------------- */
static void crop_center(const double* src, int ks, double* dst, int kt) {
    const int s0 = ks / 2 - kt / 2;
    for (int a = 0; a < kt; ++a)
        for (int b = 0; b < kt; ++b) dst[a * kt + b] = src[(s0 + a) * ks + (s0 + b)];
}

ORA_API void ora_ktransform_fwd(const float* w_max, const int* ks, int nsteps,
                                const float* const* mats, int transform, float* f, long C) {
    const int kmax = ks[0], K = ks[nsteps];
    double cur[81], crop[81], nxt[81];
    for (long c = 0; c < C; ++c) {
        for (int e = 0; e < kmax * kmax; ++e) cur[e] = (double)w_max[c * kmax * kmax + e];
        if (!transform) {
            crop_center(cur, kmax, crop, K);
            for (int e = 0; e < K * K; ++e) f[c * K * K + e] = (float)crop[e];
            continue;
        }
        int kc = kmax;
        for (int s = 0; s < nsteps; ++s) {
            const int kt = ks[s + 1], q = kt * kt;
            crop_center(cur, kc, crop, kt);
            /* the reference rounds the intermediate filter to fp32 between steps */
            for (int t = 0; t < q; ++t) {
                double a = 0.0;
                for (int u = 0; u < q; ++u) a += crop[u] * (double)mats[s][t * q + u];
                nxt[t] = (double)(float)a;
            }
            memcpy(cur, nxt, sizeof(double) * (size_t)q);
            kc = kt;
        }
        for (int e = 0; e < K * K; ++e) f[c * K * K + e] = (float)cur[e];
    }
}

/* Backward of ora_ktransform_fwd.
 *   df:     [C, K, K]
 *   dw_max: [Cmax, kmax, kmax], pre-zeroed by the caller; rows c < C receive the gradient
 *           (dense, exact zeros outside the crop window -- SURVEY.md 8a fact 1)
 *   dmats[s]: [ks[s+1]^2, ks[s+1]^2] gradient of step s's matrix (written, not accumulated);
 *           only steps 0..nsteps-1 exist -- matrices of unused steps get no gradient (None). */
ORA_API void ora_ktransform_bwd(const float* w_max, const int* ks, int nsteps,
                                const float* const* mats, int transform, const float* df,
                                float* dw_max, float* const* dmats, long C) {
    const int kmax = ks[0], K = ks[nsteps];
    if (!transform || nsteps == 0) {
        const int s0 = kmax / 2 - K / 2;
        for (long c = 0; c < C; ++c)
            for (int a = 0; a < K; ++a)
                for (int b = 0; b < K; ++b)
                    dw_max[(c * kmax + s0 + a) * kmax + s0 + b] = df[(c * K + a) * K + b];
        return;
    }
    /* accumulate matrix grads in double */
    double* dm[8];
    for (int s = 0; s < nsteps; ++s) {
        const int q = ks[s + 1] * ks[s + 1];
        dm[s] = (double*)calloc((size_t)q * q, sizeof(double));
    }
    double filt[8][81];   /* filter entering step s (size ks[s]^2), fp32-rounded like fwd */
    double crops[8][81];  /* its centre crop to ks[s+1] */
    double g[81], gc[81];
    for (long c = 0; c < C; ++c) {
        for (int e = 0; e < kmax * kmax; ++e) filt[0][e] = (double)w_max[c * kmax * kmax + e];
        for (int s = 0; s < nsteps; ++s) {
            const int kt = ks[s + 1], q = kt * kt;
            crop_center(filt[s], ks[s], crops[s], kt);
            for (int t = 0; t < q; ++t) {
                double a = 0.0;
                for (int u = 0; u < q; ++u) a += crops[s][u] * (double)mats[s][t * q + u];
                filt[s + 1][t] = (double)(float)a;
            }
        }
        for (int e = 0; e < K * K; ++e) g[e] = (double)df[c * K * K + e];
        for (int s = nsteps - 1; s >= 0; --s) {
            const int kt = ks[s + 1], q = kt * kt, kc = ks[s];
            /* g is d(out of step s) [q]; dM[t,u] += g[t]*crop[u]; dcrop[u] = sum_t g[t] M[t,u] */
            for (int t = 0; t < q; ++t)
                for (int u = 0; u < q; ++u) dm[s][t * q + u] += g[t] * crops[s][u];
            for (int u = 0; u < q; ++u) {
                double a = 0.0;
                for (int t = 0; t < q; ++t) a += g[t] * (double)mats[s][t * q + u];
                gc[u] = a;
            }
            /* scatter the crop gradient into the centre of the ks[s]-sized filter gradient */
            for (int e = 0; e < kc * kc; ++e) g[e] = 0.0;
            const int s0 = kc / 2 - kt / 2;
            for (int a = 0; a < kt; ++a)
                for (int b = 0; b < kt; ++b) g[(s0 + a) * kc + s0 + b] = gc[a * kt + b];
        }
        for (int e = 0; e < kmax * kmax; ++e) dw_max[c * kmax * kmax + e] = (float)g[e];
    }
    for (int s = 0; s < nsteps; ++s) {
        const int q = ks[s + 1] * ks[s + 1];
        for (int e = 0; e < q * q; ++e) dmats[s][e] = (float)dm[s][e];
        free(dm[s]);
    }
}

/* ------------------------------------------------------------------------------------------
 * Dense KxK convolution of the static ConvLayer (ofa/layers.py:131-151): stride 1, dilation 1,
 * groups 1, zero padding K/2, no bias.   x: [N,Cin,H,W]  w: [Cout,Cin,K,K]  y: [N,Cout,H,W]
 * ---------------------------------------------------------------------------------------- */
ORA_API void ora_conv2d_fwd(const float* x, const float* w, float* y,
                            long N, long Cin, long Cout, long H, long W, int K) {
    const int pad = K / 2;
    double* acc = (double*)malloc(sizeof(double) * (size_t)(H * W));
    for (long n = 0; n < N; ++n)
        for (long co = 0; co < Cout; ++co) {
            for (long p = 0; p < H * W; ++p) acc[p] = 0.0;
            for (long ci = 0; ci < Cin; ++ci) {
                const float* xp = x + (n * Cin + ci) * H * W;
                const float* wp = w + (co * Cin + ci) * K * K;
                for (int i = 0; i < K; ++i)
                    for (int j = 0; j < K; ++j) {
                        const double wv = (double)wp[i * K + j];
                        for (long h = 0; h < H; ++h) {
                            long hh = h + i - pad;
                            if (hh < 0 || hh >= H) continue;
                            long w0 = pad - j > 0 ? pad - j : 0;
                            long w1 = W + pad - j < W ? W + pad - j : W;
                            for (long ww = w0; ww < w1; ++ww)
                                acc[h * W + ww] += wv * (double)xp[hh * W + ww + j - pad];
                        }
                    }
            }
            float* yp = y + (n * Cout + co) * H * W;
            for (long p = 0; p < H * W; ++p) yp[p] = (float)acc[p];
        }
    free(acc);
}

/* dx[n,ci,h,w] = sum_{co,i,j} w[co,ci,i,j] * dy[n,co,h-i+pad,w-j+pad] */
ORA_API void ora_conv2d_dgrad(const float* dy, const float* w, float* dx,
                              long N, long Cin, long Cout, long H, long W, int K) {
    const int pad = K / 2;
    double* acc = (double*)malloc(sizeof(double) * (size_t)(H * W));
    for (long n = 0; n < N; ++n)
        for (long ci = 0; ci < Cin; ++ci) {
            for (long p = 0; p < H * W; ++p) acc[p] = 0.0;
            for (long co = 0; co < Cout; ++co) {
                const float* dp = dy + (n * Cout + co) * H * W;
                const float* wp = w + (co * Cin + ci) * K * K;
                for (int i = 0; i < K; ++i)
                    for (int j = 0; j < K; ++j) {
                        const double wv = (double)wp[i * K + j];
                        for (long h = 0; h < H; ++h) {
                            long hh = h - i + pad;
                            if (hh < 0 || hh >= H) continue;
                            for (long ww = 0; ww < W; ++ww) {
                                long ws = ww - j + pad;
                                if (ws < 0 || ws >= W) continue;
                                acc[h * W + ww] += wv * (double)dp[hh * W + ws];
                            }
                        }
                    }
            }
            float* xp = dx + (n * Cin + ci) * H * W;
            for (long p = 0; p < H * W; ++p) xp[p] = (float)acc[p];
        }
    free(acc);
}

/* dw[co,ci,i,j] = sum_{n,h,w} dy[n,co,h,w] * x[n,ci,h+i-pad,w+j-pad] */
ORA_API void ora_conv2d_wgrad(const float* dy, const float* x, float* dw,
                              long N, long Cin, long Cout, long H, long W, int K) {
    const int pad = K / 2;
    for (long co = 0; co < Cout; ++co)
        for (long ci = 0; ci < Cin; ++ci)
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) {
                    double s = 0.0;
                    for (long n = 0; n < N; ++n) {
                        const float* dp = dy + (n * Cout + co) * H * W;
                        const float* xp = x + (n * Cin + ci) * H * W;
                        for (long h = 0; h < H; ++h) {
                            long hh = h + i - pad;
                            if (hh < 0 || hh >= H) continue;
                            for (long ww = 0; ww < W; ++ww) {
                                long ws = ww + j - pad;
                                if (ws < 0 || ws >= W) continue;
                                s += (double)dp[h * W + ww] * (double)xp[hh * W + ws];
                            }
                        }
                    }
                    dw[((co * Cin + ci) * K + i) * K + j] = (float)s;
                }
}

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d on the first C channels (dynamic_op.py:148-167).
 *   train: batch mean / biased var per channel; y = (x-mean)/sqrt(var+eps)*gamma+beta;
 *          running_mean = (1-m)*running_mean + m*mean; running_var uses the UNBIASED var.
 *   eval : statistics come from running_mean / running_var.
 * save_mean / save_invstd (length C) are outputs in train mode (may be NULL).
 * ---------------------------------------------------------------------------------------- */
ORA_API void ora_bn_fwd(const float* x, float* y, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int training, double momentum,
                        double eps, float* save_mean, float* save_invstd,
                        long N, long C, long HW) {
    const double M = (double)(N * HW);
    for (long c = 0; c < C; ++c) {
        double mean, var;
        if (training) {
            double s = 0.0;
            for (long n = 0; n < N; ++n) {
                const float* xp = x + (n * C + c) * HW;
                for (long p = 0; p < HW; ++p) s += (double)xp[p];
            }
            mean = s / M;
            double v = 0.0;
            for (long n = 0; n < N; ++n) {
                const float* xp = x + (n * C + c) * HW;
                for (long p = 0; p < HW; ++p) {
                    double d = (double)xp[p] - mean;
                    v += d * d;
                }
            }
            var = v / M;
            if (running_mean) {
                running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
                double unb = M > 1.0 ? v / (M - 1.0) : var;
                running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
            }
        } else {
            mean = (double)running_mean[c];
            var = (double)running_var[c];
        }
        const double invstd = 1.0 / sqrt(var + eps);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
        for (long n = 0; n < N; ++n) {
            const float* xp = x + (n * C + c) * HW;
            float* yp = y + (n * C + c) * HW;
            for (long p = 0; p < HW; ++p) yp[p] = (float)(((double)xp[p] - mean) * invstd * g + b);
        }
    }
}

/* Train-mode BN backward.  dgamma/dbeta may be NULL. */
ORA_API void ora_bn_bwd_train(const float* dy, const float* x, const float* gamma, double eps,
                              float* dx, float* dgamma, float* dbeta, long N, long C, long HW) {
    const double M = (double)(N * HW);
    for (long c = 0; c < C; ++c) {
        double s = 0.0;
        for (long n = 0; n < N; ++n) {
            const float* xp = x + (n * C + c) * HW;
            for (long p = 0; p < HW; ++p) s += (double)xp[p];
        }
        const double mean = s / M;
        double v = 0.0;
        for (long n = 0; n < N; ++n) {
            const float* xp = x + (n * C + c) * HW;
            for (long p = 0; p < HW; ++p) {
                double d = (double)xp[p] - mean;
                v += d * d;
            }
        }
        const double invstd = 1.0 / sqrt(v / M + eps);
        double sdy = 0.0, sdyx = 0.0;
        for (long n = 0; n < N; ++n) {
            const float* xp = x + (n * C + c) * HW;
            const float* dp = dy + (n * C + c) * HW;
            for (long p = 0; p < HW; ++p) {
                sdy += (double)dp[p];
                sdyx += (double)dp[p] * ((double)xp[p] - mean) * invstd;
            }
        }
        if (dgamma) dgamma[c] = (float)sdyx;
        if (dbeta) dbeta[c] = (float)sdy;
        const double g = gamma ? (double)gamma[c] : 1.0;
        for (long n = 0; n < N; ++n) {
            const float* xp = x + (n * C + c) * HW;
            const float* dp = dy + (n * C + c) * HW;
            float* gx = dx + (n * C + c) * HW;
            for (long p = 0; p < HW; ++p) {
                double xh = ((double)xp[p] - mean) * invstd;
                gx[p] = (float)(g * invstd * ((double)dp[p] - sdy / M - xh * sdyx / M));
            }
        }
    }
}

ORA_API int ora_version(void) { return 1; }
