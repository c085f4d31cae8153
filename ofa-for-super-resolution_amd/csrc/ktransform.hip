// ktransform.hip -- elastic-kernel filter (centre crop + learned transform chain) for gfx950.
//
// Replaces DynamicSeparableConv2d.get_active_filter (reference
// ofa/elastic_nn/modules/dynamic_op.py:46-71) -- 6-10 tiny ATen launches (slice / contiguous /
// view / F.linear) per block per step -- with ONE launch forward and TWO backward.
//
//   f_{s+1}[c, t] = sum_u crop_{ks[s+1]}(f_s)[c, u] * M_s[t, u]         (F.linear => f . M^T)
//
// The work is tiny (<= 384 x 25 x 25 MACs): the kernel is launch-latency bound, so the design
// goal is "one launch, no host sync, no temporaries in HBM" rather than bandwidth.
// Algorithmic bytes: 4*(C*kmax^2 + sum_s q_s^2 + C*K^2).
//
// fwd:   one 64-thread block (one wave) per channel row; the chain lives in LDS.
// bwd A: same decomposition; recomputes the chain, walks it backwards, writes the dense
//        dw_max row (zeros outside the crop window) and parks per-channel (g_s, crop_s) vectors
//        in the workspace.
// bwd B: dM_s[t,u] = sum_c g_s[c,t] * crop_s[c,u]; 16 outputs x 16 channel-lanes per block,
//        fixed summation order => deterministic.
#include "ofasr_common.h"

namespace ofasr {

constexpr int KT_MAXK = 9;
constexpr int KT_MAXQ = KT_MAXK * KT_MAXK;  // 81
constexpr int KT_MAXSTEPS = 3;

struct KtParams {
    int ks[KT_MAXSTEPS + 1];
    int nsteps;
    int transform;
    const float* mats[KT_MAXSTEPS];
    float* dmats[KT_MAXSTEPS];
    int64_t ws_off[KT_MAXSTEPS];  // float offset of step s's [G | CR] block in the workspace
};

__device__ __forceinline__ int crop_index(int ksrc, int kt, int e) {
    const int s0 = ksrc / 2 - kt / 2;
    return (s0 + e / kt) * ksrc + s0 + e % kt;
}

__global__ void __launch_bounds__(64) kt_fwd_kernel(const float* __restrict__ w_max, float* __restrict__ f,
                                                    KtParams p) {
    __shared__ float buf[2][KT_MAXQ];
    const int c = blockIdx.x, t = threadIdx.x;
    const int kmax = p.ks[0], K = p.ks[p.nsteps];
    const float* wr = w_max + (int64_t)c * kmax * kmax;
    if (!p.transform || p.nsteps == 0) {
        for (int e = t; e < K * K; e += 64) f[(int64_t)c * K * K + e] = wr[crop_index(kmax, K, e)];
        return;
    }
    for (int e = t; e < kmax * kmax; e += 64) buf[0][e] = wr[e];
    __syncthreads();
    int cur = 0, kc = kmax;
    for (int s = 0; s < p.nsteps; ++s) {
        const int kt = p.ks[s + 1], q = kt * kt;
        const float* M = p.mats[s];
        for (int o = t; o < q; o += 64) {
            float a = 0.f;
            for (int u = 0; u < q; ++u) a = fmaf(buf[cur][crop_index(kc, kt, u)], M[o * q + u], a);
            buf[cur ^ 1][o] = a;
        }
        __syncthreads();
        cur ^= 1;
        kc = kt;
    }
    for (int e = t; e < K * K; e += 64) f[(int64_t)c * K * K + e] = buf[cur][e];
}

__global__ void __launch_bounds__(64) kt_bwd_chain_kernel(const float* __restrict__ w_max,
                                                          const float* __restrict__ df,
                                                          float* __restrict__ dw_max, float* __restrict__ ws,
                                                          KtParams p, int C) {
    __shared__ float filt[KT_MAXSTEPS + 1][KT_MAXQ];
    __shared__ float g[2][KT_MAXQ];
    const int c = blockIdx.x, t = threadIdx.x;
    const int kmax = p.ks[0], K = p.ks[p.nsteps];
    const float* wr = w_max + (int64_t)c * kmax * kmax;
    float* dwr = dw_max + (int64_t)c * kmax * kmax;
    if (!p.transform || p.nsteps == 0) {
        const int s0 = kmax / 2 - K / 2;
        for (int e = t; e < kmax * kmax; e += 64) {
            const int a = e / kmax - s0, b = e % kmax - s0;
            dwr[e] = (a >= 0 && a < K && b >= 0 && b < K) ? df[(int64_t)c * K * K + a * K + b] : 0.f;
        }
        return;
    }
    for (int e = t; e < kmax * kmax; e += 64) filt[0][e] = wr[e];
    __syncthreads();
    for (int s = 0; s < p.nsteps; ++s) {
        const int kt = p.ks[s + 1], q = kt * kt, kc = p.ks[s];
        const float* M = p.mats[s];
        for (int o = t; o < q; o += 64) {
            float a = 0.f;
            for (int u = 0; u < q; ++u) a = fmaf(filt[s][crop_index(kc, kt, u)], M[o * q + u], a);
            filt[s + 1][o] = a;
        }
        __syncthreads();
    }
    for (int e = t; e < K * K; e += 64) g[0][e] = df[(int64_t)c * K * K + e];
    __syncthreads();
    int cur = 0;
    for (int s = p.nsteps - 1; s >= 0; --s) {
        const int kt = p.ks[s + 1], q = kt * kt, kc = p.ks[s];
        const float* M = p.mats[s];
        float* G = ws + p.ws_off[s] + (int64_t)c * q;
        float* CR = ws + p.ws_off[s] + (int64_t)C * q + (int64_t)c * q;
        for (int e = t; e < q; e += 64) {
            G[e] = g[cur][e];
            CR[e] = filt[s][crop_index(kc, kt, e)];
        }
        // gradient w.r.t. the ks[s]-sized filter entering step s: zero outside the centre crop
        const int s0 = kc / 2 - kt / 2;
        for (int e = t; e < kc * kc; e += 64) {
            const int a = e / kc - s0, b = e % kc - s0;
            float v = 0.f;
            if (a >= 0 && a < kt && b >= 0 && b < kt) {
                const int u = a * kt + b;
                for (int o = 0; o < q; ++o) v = fmaf(g[cur][o], M[o * q + u], v);
            }
            g[cur ^ 1][e] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int e = t; e < kmax * kmax; e += 64) dwr[e] = g[cur][e];
}

// dM[t,u] = sum_c G[c,t] * CR[c,u].  block = 256 threads = 16 outputs x 16 channel lanes.
__global__ void __launch_bounds__(256) kt_bwd_mat_kernel(const float* __restrict__ G,
                                                         const float* __restrict__ CR, float* __restrict__ dM,
                                                         int q, int C) {
    const int o = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int cl = threadIdx.x & 15;
    float a = 0.f;
    if (o < q * q) {
        const int t = o / q, u = o % q;
        for (int c = cl; c < C; c += 16) a = fmaf(G[(int64_t)c * q + t], CR[(int64_t)c * q + u], a);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) a += __shfl_xor(a, off, 16);
    if (o < q * q && cl == 0) dM[o] = a;
}


// ---- batched forms (round 3): the kernel transforms of ALL blocks of an MB stack in one launch per phase.  Per step the
// per-block launches were 14 + 14 + ~15 kernels of <= 17 us (0.5 ms of kernel time, 43 launch gaps); the work per block
// is unchanged (the kernels below are the per-block kernels with a job lookup in front).
constexpr int KT_MAXJOBS = 16;
struct KtBatch {
    int n;
    int start[KT_MAXJOBS + 1];      // first workgroup of job j (prefix sums)
    const float* w_max[KT_MAXJOBS];
    float* f[KT_MAXJOBS];           // fwd: the active filter
    const float* df[KT_MAXJOBS];    // bwd
    float* dw_max[KT_MAXJOBS];
    float* ws[KT_MAXJOBS];
    int C[KT_MAXJOBS];
    KtParams p[KT_MAXJOBS];
};
struct KtMatBatch {
    int n;
    int start[2 * KT_MAXJOBS + 1];
    const float* G[2 * KT_MAXJOBS];
    const float* CR[2 * KT_MAXJOBS];
    float* dM[2 * KT_MAXJOBS];
    int q[2 * KT_MAXJOBS];
    int C[2 * KT_MAXJOBS];
};

template <typename B> __device__ __forceinline__ int kt_job_of(const B& b, int blk) {
    int j = 0;
    while (j + 1 < b.n && blk >= b.start[j + 1]) ++j;
    return j;
}

__global__ void __launch_bounds__(64) kt_fwd_batch_kernel(KtBatch b) {
    __shared__ float buf[2][KT_MAXQ];
    const int j = kt_job_of(b, blockIdx.x);
    const KtParams& p = b.p[j];
    const int c = blockIdx.x - b.start[j], t = threadIdx.x;
    const int kmax = p.ks[0], K = p.ks[p.nsteps];
    const float* wr = b.w_max[j] + (int64_t)c * kmax * kmax;
    float* f = b.f[j];
    if (!p.transform || p.nsteps == 0) {
        for (int e = t; e < K * K; e += 64) f[(int64_t)c * K * K + e] = wr[crop_index(kmax, K, e)];
        return;
    }
    for (int e = t; e < kmax * kmax; e += 64) buf[0][e] = wr[e];
    __syncthreads();
    int cur = 0, kc = kmax;
    for (int s = 0; s < p.nsteps; ++s) {
        const int kt = p.ks[s + 1], q = kt * kt;
        const float* M = p.mats[s];
        for (int o = t; o < q; o += 64) {
            float a = 0.f;
            for (int u = 0; u < q; ++u) a = fmaf(buf[cur][crop_index(kc, kt, u)], M[o * q + u], a);
            buf[cur ^ 1][o] = a;
        }
        __syncthreads();
        cur ^= 1;
        kc = kt;
    }
    for (int e = t; e < K * K; e += 64) f[(int64_t)c * K * K + e] = buf[cur][e];
}

__global__ void __launch_bounds__(64) kt_bwd_chain_batch_kernel(KtBatch b) {
    __shared__ float filt[KT_MAXSTEPS + 1][KT_MAXQ];
    __shared__ float g[2][KT_MAXQ];
    const int j = kt_job_of(b, blockIdx.x);
    const KtParams& p = b.p[j];
    const int C = b.C[j];
    const int c = blockIdx.x - b.start[j], t = threadIdx.x;
    const int kmax = p.ks[0], K = p.ks[p.nsteps];
    const float* wr = b.w_max[j] + (int64_t)c * kmax * kmax;
    const float* df = b.df[j];
    float* dwr = b.dw_max[j] + (int64_t)c * kmax * kmax;
    float* ws = b.ws[j];
    if (!p.transform || p.nsteps == 0) {
        const int s0 = kmax / 2 - K / 2;
        for (int e = t; e < kmax * kmax; e += 64) {
            const int a = e / kmax - s0, bb = e % kmax - s0;
            dwr[e] = (a >= 0 && a < K && bb >= 0 && bb < K) ? df[(int64_t)c * K * K + a * K + bb] : 0.f;
        }
        return;
    }
    for (int e = t; e < kmax * kmax; e += 64) filt[0][e] = wr[e];
    __syncthreads();
    for (int s = 0; s < p.nsteps; ++s) {
        const int kt = p.ks[s + 1], q = kt * kt, kc = p.ks[s];
        const float* M = p.mats[s];
        for (int o = t; o < q; o += 64) {
            float a = 0.f;
            for (int u = 0; u < q; ++u) a = fmaf(filt[s][crop_index(kc, kt, u)], M[o * q + u], a);
            filt[s + 1][o] = a;
        }
        __syncthreads();
    }
    for (int e = t; e < K * K; e += 64) g[0][e] = df[(int64_t)c * K * K + e];
    __syncthreads();
    int cur = 0;
    for (int s = p.nsteps - 1; s >= 0; --s) {
        const int kt = p.ks[s + 1], q = kt * kt, kc = p.ks[s];
        const float* M = p.mats[s];
        float* G = ws + p.ws_off[s] + (int64_t)c * q;
        float* CR = ws + p.ws_off[s] + (int64_t)C * q + (int64_t)c * q;
        for (int e = t; e < q; e += 64) {
            G[e] = g[cur][e];
            CR[e] = filt[s][crop_index(kc, kt, e)];
        }
        const int s0 = kc / 2 - kt / 2;
        for (int e = t; e < kc * kc; e += 64) {
            const int a = e / kc - s0, bb = e % kc - s0;
            float v = 0.f;
            if (a >= 0 && a < kt && bb >= 0 && bb < kt) {
                const int u = a * kt + bb;
                for (int o = 0; o < q; ++o) v = fmaf(g[cur][o], M[o * q + u], v);
            }
            g[cur ^ 1][e] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int e = t; e < kmax * kmax; e += 64) dwr[e] = g[cur][e];
}

__global__ void __launch_bounds__(256) kt_bwd_mat_batch_kernel(KtMatBatch b) {
    const int j = kt_job_of(b, blockIdx.x);
    const int q = b.q[j], C = b.C[j];
    const float* G = b.G[j];
    const float* CR = b.CR[j];
    const int o = (blockIdx.x - b.start[j]) * 16 + (threadIdx.x >> 4);
    const int cl = threadIdx.x & 15;
    float a = 0.f;
    if (o < q * q) {
        const int t = o / q, u = o % q;
        for (int c = cl; c < C; c += 16) a = fmaf(G[(int64_t)c * q + t], CR[(int64_t)c * q + u], a);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) a += __shfl_xor(a, off, 16);
    if (o < q * q && cl == 0) b.dM[j][o] = a;
}

static int fill_params(const char* name, KtParams& p, const int* ks, int nsteps, const float* const* mats,
                       int transform, int64_t C) {
    OFASR_REQUIRE(ks != nullptr, OFASR_ERR_INVALID_ARG, "%s: ks is null", name);
    OFASR_REQUIRE(nsteps >= 0 && nsteps <= KT_MAXSTEPS, OFASR_ERR_UNSUPPORTED, "%s: nsteps=%d not in [0,%d]",
                  name, nsteps, KT_MAXSTEPS);
    OFASR_REQUIRE(C >= 0 && C <= INT32_MAX, OFASR_ERR_INVALID_ARG, "%s: bad C=%lld", name, (long long)C);
    for (int s = 0; s <= nsteps; ++s) {
        OFASR_REQUIRE(ks[s] >= 1 && ks[s] <= KT_MAXK && (ks[s] & 1), OFASR_ERR_UNSUPPORTED,
                      "%s: kernel size %d must be odd and <= %d", name, ks[s], KT_MAXK);
        OFASR_REQUIRE(s == 0 || ks[s] < ks[s - 1], OFASR_ERR_INVALID_ARG, "%s: ks must strictly decrease", name);
        p.ks[s] = ks[s];
    }
    p.nsteps = nsteps;
    p.transform = transform ? 1 : 0;
    for (int s = 0; s < KT_MAXSTEPS; ++s) {
        p.mats[s] = nullptr;
        p.dmats[s] = nullptr;
        p.ws_off[s] = 0;
    }
    if (p.transform)
        for (int s = 0; s < nsteps; ++s) {
            OFASR_REQUIRE(mats && mats[s], OFASR_ERR_INVALID_ARG, "%s: mats[%d] is null", name, s);
            p.mats[s] = mats[s];
        }
    return OFASR_OK;
}


int ktransform_fwd_batch(const KtJob* jobs, int n, void* stream) {
    const char* name = "ktransform_fwd_batch";
    OFASR_REQUIRE(jobs && n > 0 && n <= KT_MAXJOBS, OFASR_ERR_INVALID_ARG, "%s: bad job list", name);
    KtBatch b{};
    b.n = n;
    int blocks = 0;
    for (int j = 0; j < n; ++j) {
        int rc = fill_params(name, b.p[j], jobs[j].ks, jobs[j].nsteps, jobs[j].mats, jobs[j].transform, jobs[j].C);
        if (rc) return rc;
        OFASR_REQUIRE(jobs[j].w_max && jobs[j].f && jobs[j].C > 0, OFASR_ERR_INVALID_ARG, "%s: job %d incomplete", name, j);
        b.start[j] = blocks;
        b.w_max[j] = jobs[j].w_max;
        b.f[j] = jobs[j].f;
        b.C[j] = (int)jobs[j].C;
        blocks += (int)jobs[j].C;
    }
    b.start[n] = blocks;
    OFASR_LAUNCH(kt_fwd_batch_kernel, dim3((unsigned)blocks), dim3(64), 0, as_stream(stream), b);
    return check_launch(name);
}

int ktransform_bwd_batch(const KtJob* jobs, int n, void* stream) {
    const char* name = "ktransform_bwd_batch";
    OFASR_REQUIRE(jobs && n > 0 && n <= KT_MAXJOBS, OFASR_ERR_INVALID_ARG, "%s: bad job list", name);
    KtBatch b{};
    KtMatBatch m{};
    b.n = n;
    int blocks = 0, mblocks = 0;
    for (int j = 0; j < n; ++j) {
        const KtJob& jb = jobs[j];
        int rc = fill_params(name, b.p[j], jb.ks, jb.nsteps, jb.mats, jb.transform, jb.C);
        if (rc) return rc;
        OFASR_REQUIRE(jb.w_max && jb.df && jb.dw_max && jb.C > 0, OFASR_ERR_INVALID_ARG, "%s: job %d incomplete", name, j);
        const bool chain = b.p[j].transform && jb.nsteps > 0;
        if (chain) {
            const size_t need = ofasr_ktransform_bwd_workspace(jb.ks, jb.nsteps, jb.C);
            OFASR_REQUIRE(jb.ws && jb.ws_bytes >= need, OFASR_ERR_WORKSPACE, "%s: job %d workspace %zu B < required %zu B", name, j,
                          jb.ws_bytes, need);
            int64_t off = 0;
            for (int s = 0; s < jb.nsteps; ++s) {
                OFASR_REQUIRE(jb.dmats && jb.dmats[s], OFASR_ERR_INVALID_ARG, "%s: job %d dmats[%d] is null", name, j, s);
                b.p[j].dmats[s] = jb.dmats[s];
                b.p[j].ws_off[s] = off;
                const int q = jb.ks[s + 1] * jb.ks[s + 1];
                OFASR_REQUIRE(m.n < 2 * KT_MAXJOBS, OFASR_ERR_UNSUPPORTED, "%s: too many transform steps", name);
                m.start[m.n] = mblocks;
                m.G[m.n] = (const float*)jb.ws + off;
                m.CR[m.n] = (const float*)jb.ws + off + (int64_t)jb.C * q;
                m.dM[m.n] = jb.dmats[s];
                m.q[m.n] = q;
                m.C[m.n] = (int)jb.C;
                mblocks += (int)cdiv((int64_t)q * q, 16);
                ++m.n;
                off += (int64_t)2 * jb.C * q;
            }
        }
        b.start[j] = blocks;
        b.w_max[j] = jb.w_max;
        b.df[j] = jb.df;
        b.dw_max[j] = jb.dw_max;
        b.ws[j] = (float*)jb.ws;
        b.C[j] = (int)jb.C;
        blocks += (int)jb.C;
    }
    b.start[n] = blocks;
    hipStream_t st = as_stream(stream);
    OFASR_LAUNCH(kt_bwd_chain_batch_kernel, dim3((unsigned)blocks), dim3(64), 0, st, b);
    int rc = check_launch(name);
    if (rc || m.n == 0) return rc;
    m.start[m.n] = mblocks;
    OFASR_LAUNCH(kt_bwd_mat_batch_kernel, dim3((unsigned)mblocks), dim3(256), 0, st, m);
    return check_launch(name);
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT int ofasr_ktransform_fwd(const float* w_max, const int* ks, int nsteps, const float* const* mats,
                                      int transform, float* f, int64_t C, void* stream) {
    KtParams p;
    int rc = fill_params("ofasr_ktransform_fwd", p, ks, nsteps, mats, transform, C);
    if (rc) return rc;
    OFASR_REQUIRE(w_max && f, OFASR_ERR_INVALID_ARG, "ofasr_ktransform_fwd: null pointer");
    if (C == 0) return OFASR_OK;
    OFASR_LAUNCH(kt_fwd_kernel, dim3((unsigned)C), dim3(64), 0, as_stream(stream), w_max, f, p);
    return check_launch("ofasr_ktransform_fwd");
}

OFASR_EXPORT size_t ofasr_ktransform_bwd_workspace(const int* ks, int nsteps, int64_t C) {
    if (!ks || nsteps <= 0 || nsteps > KT_MAXSTEPS || C <= 0) return 0;
    size_t fl = 0;
    for (int s = 0; s < nsteps; ++s) fl += (size_t)2 * (size_t)C * (size_t)(ks[s + 1] * ks[s + 1]);
    return fl * sizeof(float);
}

OFASR_EXPORT int ofasr_ktransform_bwd(const float* w_max, const int* ks, int nsteps, const float* const* mats,
                                      int transform, const float* df, float* dw_max, float* const* dmats,
                                      int64_t C, void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_ktransform_bwd";
    KtParams p;
    int rc = fill_params(name, p, ks, nsteps, mats, transform, C);
    if (rc) return rc;
    OFASR_REQUIRE(w_max && df && dw_max, OFASR_ERR_INVALID_ARG, "%s: null pointer", name);
    if (C == 0) return OFASR_OK;
    const bool chain = p.transform && nsteps > 0;
    if (chain) {
        const size_t need = ofasr_ktransform_bwd_workspace(ks, nsteps, C);
        OFASR_REQUIRE(workspace && workspace_bytes >= need, OFASR_ERR_WORKSPACE,
                      "%s: workspace %zu B < required %zu B", name, workspace_bytes, need);
        int64_t off = 0;
        for (int s = 0; s < nsteps; ++s) {
            OFASR_REQUIRE(dmats && dmats[s], OFASR_ERR_INVALID_ARG, "%s: dmats[%d] is null", name, s);
            p.dmats[s] = dmats[s];
            p.ws_off[s] = off;
            off += (int64_t)2 * C * ks[s + 1] * ks[s + 1];
        }
    }
    hipStream_t st = as_stream(stream);
    OFASR_LAUNCH(kt_bwd_chain_kernel, dim3((unsigned)C), dim3(64), 0, st, w_max, df, dw_max,
                       (float*)workspace, p, (int)C);
    rc = check_launch(name);
    if (rc || !chain) return rc;
    for (int s = 0; s < nsteps; ++s) {
        const int q = ks[s + 1] * ks[s + 1];
        const float* G = (const float*)workspace + p.ws_off[s];
        const float* CR = G + (int64_t)C * q;
        OFASR_LAUNCH(kt_bwd_mat_kernel, dim3((unsigned)cdiv((int64_t)q * q, 16)), dim3(256), 0, st, G, CR,
                           p.dmats[s], q, (int)C);
        rc = check_launch(name);
        if (rc) return rc;
    }
    return OFASR_OK;
}
