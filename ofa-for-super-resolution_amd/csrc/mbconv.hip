// mbconv.hip -- composite MB-block entry points: one host call enqueues every kernel of a
// DynamicMBConvLayer (+ identity shortcut) forward or backward (see include/ofasr.h).
//
// No new arithmetic lives here: it sequences the kernels of pwconv.hip / dwconv.hip / ktransform.hip /
// bnact.hip on one stream, carving a caller-provided workspace.  With 16-bit activations and vector-eligible
// shapes BN1/BN2 + ReLU6 are not separate passes: the depthwise / project kernels (and their weight-gradient
// kernels) apply scale/shift/clamp to the pre-BN tensor as they read it (InputXf), so the activated mid tensors
// are never written or re-read.  The point is host cost: per block the
// Python side makes 1 FFI call per direction instead of ~25 (each with its own autograd node, allocations
// and ctypes marshalling), which is what bounded the training step once the kernels were fast.
#include "ofasr_common.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace ofasr {

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static bool bn_fold_enabled() {
    static const bool on = [] { const char* e = getenv("OFASR_MBCONV_BN_BWD_FOLD"); return !(e && e[0] == '0'); }();
    return on;
}
// set by ofasr_mbstack_fwd / _bwd around their per-block calls: the stack runs the kernel transforms of all blocks in one
// launch per phase itself (bit 0: the active filter is already in stat_buf; bit 1: leave the filter gradient's chain to the
// caller)
static thread_local int t_skip_kt = 0;
static std::atomic<int> g_bn_bwd_stat{[] { const char* e = getenv("OFASR_MBCONV_BN_BWD_STAT"); return (e && e[0] == '1') ? 1 : 0; }()};

struct MbSizes {
    size_t es;            // activation element size
    size_t mid_elems;     // N*mid*HW
    size_t out_elems;     // N*Cout*HW
    size_t ws_bn;         // statistics partials (largest BN)
    size_t ws_bn_bwd;
    size_t ws_dw;
    size_t ws_pw;
    size_t ws_kt;
    size_t scratch;       // main-stream region shared by whichever BN kernel is running
    size_t stat_a;        // bytes of the first statistics-partials region inside it (BN1 / BN3); BN2's follows
    size_t side;          // side-stream region shared by whichever weight-gradient kernel is running
    size_t df_bytes;      // depthwise filter gradient (lives from dw wgrad to the kernel-transform backward)
    size_t coef_off;      // BN1 / BN2 backward coefficients (ka | kbi each, mid floats): read by both streams
    size_t total;
};

static MbSizes mb_sizes(const ofasr_mbconv_desc* d) {
    MbSizes s;
    const int64_t HW = d->H * d->W;
    s.es = d->dtype == OFASR_F32 ? 4 : 2;
    s.mid_elems = (size_t)(d->N * d->mid * HW);
    s.out_elems = (size_t)(d->N * d->Cout * HW);
    const int64_t cbig = d->mid > d->Cout ? d->mid : d->Cout;
    s.ws_bn = align_up(ofasr_bn_workspace(d->N, cbig), 256);
    s.ws_bn_bwd = align_up(ofasr_bn_act_bwd_workspace(d->N, cbig), 256);
    s.ws_dw = align_up(ofasr_dwconv_wgrad_workspace(d->N, d->mid, d->H, d->W, d->K), 256);
    size_t pw1 = ofasr_pwconv_wgrad_workspace(d->N, d->Cin, d->mid, HW);
    size_t pw2 = ofasr_pwconv_wgrad_workspace(d->N, d->mid, d->Cout, HW);
    s.ws_pw = align_up(pw1 > pw2 ? pw1 : pw2, 256);
    s.ws_kt = align_up(ofasr_ktransform_bwd_workspace(d->ks, d->chain_len - 1, d->mid), 256);
    // main stream: BN statistics / BN backward partials (kernels run back to back, so one slab for the largest user);
    // side stream (weight-gradient kernels of the backward, ofasr_mbconv_bwd): its own slab + the depthwise filter
    // gradient + the kernel-transform workspace, so the two streams never share scratch
    s.scratch = s.ws_bn > s.ws_bn_bwd ? s.ws_bn : s.ws_bn_bwd;
    {   // statistics partials written by the conv kernels themselves (fused forward): [C][units] float2
        const size_t u1 = (size_t)pwconv_stat_units(d->N, d->Cin, HW) * (size_t)d->mid;
        const size_t u2 = (size_t)dwconv_stat_units(d->N, d->H, d->W, d->K, d->dtype) * (size_t)d->mid;
        const size_t u3 = (size_t)pwconv_stat_units(d->N, d->mid, HW) * (size_t)d->Cout;
        // two regions: a consumer that folds its input's partials (region B) writes its own output's partials (region A)
        s.stat_a = align_up((u1 > u3 ? u1 : u3) * sizeof(float2), 256);
        const size_t ws_stat = s.stat_a + align_up(u2 * sizeof(float2), 256);
        if (ws_stat > s.scratch) s.scratch = ws_stat;
    }
    s.side = s.ws_dw > s.ws_pw ? s.ws_dw : s.ws_pw;
    s.df_bytes = align_up((size_t)(d->mid * d->K * d->K) * sizeof(float), 256);
    s.coef_off = s.scratch + s.side + s.df_bytes + s.ws_kt + 256;
    s.total = s.coef_off + align_up((size_t)(4 * d->mid) * sizeof(float), 256);
    return s;
}

static int check_desc(const char* name, const ofasr_mbconv_desc* d) {
    OFASR_REQUIRE(d != nullptr, OFASR_ERR_INVALID_ARG, "%s: null descriptor", name);
    OFASR_REQUIRE(d->N > 0 && d->Cin > 0 && d->mid > 0 && d->Cout > 0 && d->H > 0 && d->W > 0, OFASR_ERR_INVALID_ARG,
                  "%s: bad shape", name);
    OFASR_REQUIRE(d->chain_len >= 1 && d->chain_len <= 4, OFASR_ERR_INVALID_ARG, "%s: bad kernel chain", name);
    OFASR_REQUIRE(d->ks[d->chain_len - 1] == d->K, OFASR_ERR_INVALID_ARG, "%s: chain does not end at K", name);
    OFASR_REQUIRE(d->w1 && d->w2 && d->wdw_max, OFASR_ERR_INVALID_ARG, "%s: null weight", name);
    OFASR_REQUIRE(!d->residual || d->Cin == d->Cout, OFASR_ERR_INVALID_ARG, "%s: residual needs Cin == Cout", name);
    for (int i = 0; i < 3; ++i)
        OFASR_REQUIRE(d->gamma[i] && d->beta[i] && d->running_mean[i] && d->running_var[i], OFASR_ERR_INVALID_ARG,
                      "%s: null BN tensor %d", name, i);
    return OFASR_OK;
}

__global__ void bump_counters_kernel(int64_t* a, int64_t* b, int64_t* c) {
    if (threadIdx.x == 0) {
        if (a) *a += 1;
        if (b) *b += 1;
        if (c) *c += 1;
    }
}

// One non-blocking side stream + fork/join events per process (one process drives one GPU): the weight-gradient
// kernels of a block's backward are independent of the input-gradient chain, so they run beside it and fill the
// load / drain phases in which a lone bandwidth-bound kernel leaves HBM idle.  Created on first use; legal inside
// hipGraph capture (fork/join through events).  OFASR_MBCONV_SIDE_STREAM=0 keeps everything on the caller's stream.
//
// Deferred join (ofasr_mbconv_defer_join(1)): by default a backward call ends by making the caller's stream wait for
// the side stream, so every gradient is final in stream order when the call returns -- and the next block's
// input-gradient chain queues behind this block's weight gradients (per block the cost is max(main, side), 7 % of the
// north-star step).  With the join deferred the side stream runs free: the call returns with dx (and the BN gradients,
// which the main chain writes) final in stream order, the weight / transform-matrix gradients only after
// ofasr_mbconv_join(stream).  Until then the caller keeps every buffer of the call alive; a later call whose tmp_buf
// or workspace overlaps those of an unjoined one waits for that one's side work first.
struct PendingSide {
    const char *t0, *t1, *w0, *w1;   // tmp_buf and workspace byte ranges the side kernels of the call still use
    hipEvent_t done;                  // recorded on the side stream after the call's last side kernel
};
constexpr size_t MAX_PENDING = 64;
struct SideStream {
    bool ready = false, enabled = true, defer = false;
    hipStream_t s = nullptr;
    hipEvent_t fork[3] = {nullptr, nullptr, nullptr};
    hipEvent_t join = nullptr;
    std::mutex mu;                    // guards pending / pool / defer
    std::vector<PendingSide> pending;
    std::vector<hipEvent_t> pool;     // done-events, created on demand, reused after a join
};
static SideStream& side_stream() {
    static SideStream ss;
    if (!ss.ready) {
        ss.ready = true;
        const char* e = getenv("OFASR_MBCONV_SIDE_STREAM");
        ss.enabled = !(e && e[0] == '0');
        if (ss.enabled) {
            // default priority.  OFASR_SIDE_STREAM_PRIORITY=low|high for experiments: measured on the north-star step,
            // low 2160-2200 and high 2280 against 2275 images/s at the default -- the input-gradient chain is
            // the critical path, yet starving the weight gradients only moves the wait to the end of backward
            const char* pe = getenv("OFASR_SIDE_STREAM_PRIORITY");
            int least = 0, greatest = 0;
            bool ok;
            if (pe && (pe[0] == 'l' || pe[0] == 'h') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
                least != greatest)
                ok = hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, pe[0] == 'l' ? least : greatest) == hipSuccess;
            else
                ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreateWithFlags(&ss.fork[i], hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
            ss.enabled = ok;
        }
    }
    return ss;
}

// `st` waits for everything enqueued on the side stream so far; clears the pending list (ss.mu held by the caller)
static int join_side_locked(SideStream& ss, hipStream_t st, const char* name) {
    if (!ss.enabled) return OFASR_OK;
    hipError_t e = hipEventRecord(ss.join, ss.s);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, ss.join, 0);
    OFASR_REQUIRE(e == hipSuccess, OFASR_ERR_LAUNCH, "%s: stream join failed: %s", name, hipGetErrorString(e));
    ss.pending.clear();
    return OFASR_OK;
}

struct StatView {
    float *mean, *invstd, *scale, *shift;
};
static StatView stat_view(float* base, int which, int64_t mid, int64_t cout) {
    const int64_t off[3] = {0, 4 * mid, 8 * mid};
    const int64_t C = which < 2 ? mid : cout;
    float* p = base + off[which];
    return StatView{p, p + C, p + 2 * C, p + 3 * C};
}

// BatchNorm(+act)(+residual) forward of one of the block's three BNs
static int bn_forward(const ofasr_mbconv_desc* d, int which, const void* x, const void* res, void* y, int64_t C,
                      int act, float* stat_buf, void* ws, size_t ws_bytes, void* stream) {
    const int64_t HW = d->H * d->W;
    StatView sv = stat_view(stat_buf, which, d->mid, d->Cout);
    return ofasr_bn_fwd(x, res, y, d->gamma[which], d->beta[which], d->running_mean[which], d->running_var[which],
                        d->bn_momentum[which], d->bn_eps[which], d->bn_training[which], sv.mean, d->N, C, HW, act,
                        d->dtype, ws, ws_bytes, stream);
}

static InputXf xf_of(const float* stat_buf, int which, int64_t mid, int64_t cout) {
    StatView sv = stat_view(const_cast<float*>(stat_buf), which, mid, cout);
    return InputXf{sv.scale, sv.shift, sv.mean};
}

// BN1/BN2 + ReLU6 are applied by the consumer's loads (a1, a2 are never written) when the vector depthwise and the
// aligned 16-bit pointwise kernels apply to the block; forward and backward take the same decision from the
// descriptor and the (caller-provided) buffer addresses.
// The decision is a function of the descriptor alone (act_buf must then be 16-byte aligned), because it also fixes the
// layout of act_buf: the fused path keeps y1 | y2 | y3 | out only (the activated tensors a1, a2 are never written, so
// they get no room: 2*mid + 2*Cout instead of 4*mid + 2*Cout channels per pixel held for the backward).
struct ActLayout {
    bool fused;
    size_t y1, a1, y2, a2, y3, out, total;   // element offsets; a1 / a2 are meaningless when fused
};
static ActLayout act_layout(const ofasr_mbconv_desc* d, const MbSizes& s) {
    ActLayout L;
    const char* base = reinterpret_cast<const char*>(uintptr_t(4096));   // alignment of a conforming act_buf
    L.fused = dwconv_xf_supported(base, base + s.mid_elems * s.es, d->H, d->W, d->K, d->dtype) &&
              pwconv_xf_supported(base + s.mid_elems * s.es, base + 2 * s.mid_elems * s.es, d->H * d->W, d->dtype);
    if (L.fused) {
        L.y1 = 0; L.y2 = s.mid_elems; L.y3 = 2 * s.mid_elems; L.out = 2 * s.mid_elems + s.out_elems;
        L.a1 = L.a2 = 0;
    } else {
        L.y1 = 0; L.a1 = s.mid_elems; L.y2 = 2 * s.mid_elems; L.a2 = 3 * s.mid_elems; L.y3 = 4 * s.mid_elems;
        L.out = 4 * s.mid_elems + s.out_elems;
    }
    L.total = L.out + s.out_elems;
    return L;
}

}  // namespace ofasr

using namespace ofasr;

OFASR_EXPORT size_t ofasr_mbconv_workspace(const ofasr_mbconv_desc* d) {
    if (!d || d->N <= 0) return 0;
    return mb_sizes(d).total;
}

OFASR_EXPORT size_t ofasr_mbconv_act_elems(const ofasr_mbconv_desc* d) {
    if (!d || d->N <= 0) return 0;
    return act_layout(d, mb_sizes(d)).total;
}

OFASR_EXPORT size_t ofasr_mbconv_stat_floats(const ofasr_mbconv_desc* d) {
    if (!d) return 0;
    return (size_t)(8 * d->mid + 4 * d->Cout + d->mid * d->K * d->K);
}

OFASR_EXPORT int ofasr_mbconv_fwd(const ofasr_mbconv_desc* d, const void* x, void* act_buf, float* stat_buf,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    const char* name = "ofasr_mbconv_fwd";
    int rc = check_desc(name, d);
    if (rc) return rc;
    OFASR_REQUIRE(x && act_buf && stat_buf, OFASR_ERR_INVALID_ARG, "%s: null buffer", name);
    const MbSizes s = mb_sizes(d);
    OFASR_REQUIRE(workspace && workspace_bytes >= s.total, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, s.total);
    const int64_t HW = d->H * d->W;
    char* a = (char*)act_buf;
    const ActLayout lay = act_layout(d, s);
    OFASR_REQUIRE(!lay.fused || (reinterpret_cast<uintptr_t>(act_buf) & 15) == 0, OFASR_ERR_INVALID_ARG,
                  "%s: act_buf must be 16-byte aligned", name);
    void* y1 = a + lay.y1 * s.es;
    void* a1 = a + lay.a1 * s.es;
    void* y2 = a + lay.y2 * s.es;
    void* a2 = a + lay.a2 * s.es;
    void* y3 = a + lay.y3 * s.es;
    void* out = a + lay.out * s.es;
    float* f = stat_buf + 8 * d->mid + 4 * d->Cout;

    const bool fused = lay.fused;
    if (!fused && ((d->bn_training[0] && d->num_batches_tracked[0]) || (d->bn_training[1] && d->num_batches_tracked[1]) ||
                   (d->bn_training[2] && d->num_batches_tracked[2]))) {
        OFASR_LAUNCH(bump_counters_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                           d->bn_training[0] ? d->num_batches_tracked[0] : nullptr,
                           d->bn_training[1] ? d->num_batches_tracked[1] : nullptr,
                           d->bn_training[2] ? d->num_batches_tracked[2] : nullptr);
        rc = check_launch(name);
        if (rc) return rc;
    }
    // expand 1x1 -> BN + ReLU6
    // statistics in the pointwise kernels' epilogue cost almost what the statistics pass they replace costs (a
    // 5-round cross-lane reduce at the tail of a kernel whose blocks all run in phase): +0.8 % on the step, kept behind
    // a switch (OFASR_PW_EPILOGUE_STATS=0 goes back to the pass); the depthwise kernel's statistics are free (one wave
    // owns a whole plane).
    static const bool pw_stat = [] { const char* e = getenv("OFASR_PW_EPILOGUE_STATS"); return !(e && e[0] == '0'); }();
    if (pw_stat && d->bn_training[0])   // fused or not: the un-fused path folds the partials in its BN apply (below)
        rc = pwconv_fwd_stat(x, d->w1, d->ldw1, y1, d->N, d->Cin, d->mid, HW, d->dtype,
                             StatOut{(float2*)workspace, pwconv_stat_units(d->N, d->Cin, HW)}, stream);
    else
        rc = ofasr_pwconv_fwd(x, d->w1, d->ldw1, y1, d->N, d->Cin, d->mid, HW, d->dtype, stream);
    if (rc) return rc;
    if (fused) {
        // the conv kernels leave their output's per-channel (sum, sum of squares) partials in the epilogue; a one-wave-
        // per-channel finalize turns them into scale/shift (+ running statistics, + the three counters): no pass over
        // the tensors for statistics, no pass for BN1/BN2 + ReLU6 (applied by the consumer's loads)
        const double count = (double)d->N * (double)HW;
        float2* part = (float2*)workspace;                              // BN1 and BN3 partials
        float2* part_b = (float2*)((char*)workspace + s.stat_a);       // BN2 partials
        auto finalize = [&](int which, int64_t C, int P, bool bump) -> int {
            StatView sv = stat_view(stat_buf, which, d->mid, d->Cout);
            int64_t* k[3] = {nullptr, nullptr, nullptr};
            if (bump)
                for (int i = 0; i < 3; ++i) k[i] = d->bn_training[i] ? d->num_batches_tracked[i] : nullptr;
            return bn_finalize_cp(part, P, C, count, d->gamma[which], d->beta[which], d->running_mean[which],
                                  d->running_var[which], d->bn_momentum[which], d->bn_eps[which], d->bn_training[which],
                                  sv.mean, sv.invstd, sv.scale, sv.shift, k[0], k[1], k[2], stream);
        };
        const int P1 = pwconv_stat_units(d->N, d->Cin, HW);
        const int P2 = dwconv_stat_units(d->N, d->H, d->W, d->K, d->dtype);
        const int P3 = pwconv_stat_units(d->N, d->mid, HW);
        // statistics by a pass over the tensor (the pointwise default): fp64 partial slabs + the classic finalize
        auto pass_stats = [&](int which, const void* t, int64_t C, bool bump) -> int {
            StatView sv = stat_view(stat_buf, which, d->mid, d->Cout);
            if (d->bn_training[which]) {
                int r2 = ofasr_bn_stats(t, d->N, C, HW, d->dtype, workspace, s.scratch, stream);
                if (r2) return r2;
            }
            int64_t* k[3] = {nullptr, nullptr, nullptr};
            if (bump)
                for (int i = 0; i < 3; ++i) k[i] = d->bn_training[i] ? d->num_batches_tracked[i] : nullptr;
            return bn_finalize_bump(workspace, ofasr_bn_partials(d->N, C), C, count, d->gamma[which], d->beta[which],
                                    d->running_mean[which], d->running_var[which], d->bn_momentum[which],
                                    d->bn_eps[which], d->bn_training[which], sv.mean, sv.invstd, sv.scale, sv.shift,
                                    k[0], k[1], k[2], stream);
        };
        const bool fold1 = pw_stat && d->bn_training[0];   // BN1's finalize folded into the depthwise kernel's waves
        if (!fold1) {
            rc = pw_stat ? finalize(0, d->mid, P1, true) : pass_stats(0, y1, d->mid, true);
            if (rc) return rc;
        }
        rc = (t_skip_kt & 1) ? OFASR_OK
                             : ofasr_ktransform_fwd(d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, f, d->mid, stream);
        if (rc) return rc;
        {
            StatView sv = stat_view(stat_buf, 0, d->mid, d->Cout);
            BnFold f1{};
            if (fold1) {
                f1 = BnFold{part, P1, count, d->bn_momentum[0], d->bn_eps[0], d->gamma[0], d->beta[0], d->running_mean[0],
                            d->running_var[0], sv.mean, sv.invstd, sv.scale, sv.shift, {nullptr, nullptr, nullptr}};
                for (int i = 0; i < 3; ++i) f1.counters[i] = d->bn_training[i] ? d->num_batches_tracked[i] : nullptr;
            }
            rc = dwconv_fwd_xf(y1, f, y2, d->N, d->mid, d->H, d->W, d->K, d->dtype,
                               fold1 ? InputXf{nullptr, nullptr, nullptr} : xf_of(stat_buf, 0, d->mid, d->Cout), stream,
                               StatOut{d->bn_training[1] ? part_b : nullptr, P2}, f1);
        }
        if (rc) return rc;
        const StatOut so3{(pw_stat && d->bn_training[2]) ? part : nullptr, P3};
        if (d->bn_training[1] && d->Cout % 4 == 0 && pwconv_fold_supported(y2, y3, d->w2, d->ldw2, d->mid, HW, d->dtype)) {
            // BN2's finalize is folded into the project kernel's blocks (16 partials per channel)
            StatView sv = stat_view(stat_buf, 1, d->mid, d->Cout);
            rc = pwconv_fwd_fold(y2, d->w2, d->ldw2, y3, d->N, d->mid, d->Cout, HW, d->dtype,
                                 BnFold{part_b, P2, count, d->bn_momentum[1], d->bn_eps[1], d->gamma[1], d->beta[1],
                                        d->running_mean[1], d->running_var[1], sv.mean, sv.invstd, sv.scale, sv.shift,
                                        {nullptr, nullptr, nullptr}},
                                 stream, so3);
        } else {
            float2* keep = part;
            part = part_b;
            rc = finalize(1, d->mid, P2, false);
            part = keep;
            if (rc) return rc;
            rc = pwconv_fwd_xf(y2, d->w2, d->ldw2, y3, d->N, d->mid, d->Cout, HW, d->dtype,
                               xf_of(stat_buf, 1, d->mid, d->Cout), stream, so3);
        }
        if (rc) return rc;
        StatView s3 = stat_view(stat_buf, 2, d->mid, d->Cout);
        if (pw_stat)   // BN3's finalize is folded into the apply kernel's blocks
            return bn_fwd_cp(y3, d->residual ? x : nullptr, out, part, P3, d->gamma[2], d->beta[2], d->running_mean[2],
                             d->running_var[2], d->bn_momentum[2], d->bn_eps[2], d->bn_training[2], s3.mean, d->N,
                             d->Cout, HW, 0, d->dtype, stream);
        rc = pass_stats(2, y3, d->Cout, false);
        if (rc) return rc;
        return ofasr_bn_act_fwd(y3, d->residual ? x : nullptr, out, s3.scale, s3.shift, s3.mean, d->N, d->Cout, HW, 0,
                                d->dtype, stream);
    }
    // Un-fused path (fp32 activations, or shapes the fused reads do not cover).  The 1x1 convs still leave their output's
    // statistics partials in the epilogue (the generic kernels take a StatOut for every element type) and the BN apply
    // folds them in its blocks (bn_fwd_cp): no statistics pass over y1 / y3 (round 3: 28 + 14 bn_stats launches off the
    // fp32 step).
    const bool cp1 = pw_stat && d->bn_training[0], cp3 = pw_stat && d->bn_training[2];
    if (cp1) {
        const int P1 = pwconv_stat_units(d->N, d->Cin, HW);
        StatView sv = stat_view(stat_buf, 0, d->mid, d->Cout);
        rc = bn_fwd_cp(y1, nullptr, a1, (const float2*)workspace, P1, d->gamma[0], d->beta[0], d->running_mean[0],
                       d->running_var[0], d->bn_momentum[0], d->bn_eps[0], d->bn_training[0], sv.mean, d->N, d->mid, HW, 1,
                       d->dtype, stream);
    } else {
        rc = bn_forward(d, 0, y1, nullptr, a1, d->mid, 1, stat_buf, workspace, workspace_bytes, stream);
    }
    if (rc) return rc;
    // active depthwise filter -> depthwise -> BN + ReLU6
    rc = (t_skip_kt & 1) ? OFASR_OK : ofasr_ktransform_fwd(d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, f, d->mid,
                              stream);
    if (rc) return rc;
    const int P2u = dwconv_stat_units(d->N, d->H, d->W, d->K, d->dtype);
    if (pw_stat && d->bn_training[1] && P2u > 0 && dwconv_stat_supported(a1, y2, d->H, d->W, d->K, d->dtype)) {
        // the depthwise kernel's statistics are free (a wave owns whole rows of one plane): BN2 without its statistics pass
        rc = dwconv_fwd_stat(a1, f, y2, d->N, d->mid, d->H, d->W, d->K, d->dtype, StatOut{(float2*)workspace, P2u}, stream);
        if (rc) return rc;
        StatView sv = stat_view(stat_buf, 1, d->mid, d->Cout);
        rc = bn_fwd_cp(y2, nullptr, a2, (const float2*)workspace, P2u, d->gamma[1], d->beta[1], d->running_mean[1],
                       d->running_var[1], d->bn_momentum[1], d->bn_eps[1], d->bn_training[1], sv.mean, d->N, d->mid, HW, 1,
                       d->dtype, stream);
    } else {
        rc = ofasr_dwconv_fwd(a1, f, y2, d->N, d->mid, d->H, d->W, d->K, d->dtype, stream);
        if (rc) return rc;
        rc = bn_forward(d, 1, y2, nullptr, a2, d->mid, 1, stat_buf, workspace, workspace_bytes, stream);
    }
    if (rc) return rc;
    // project 1x1 -> BN (+ shortcut)
    if (cp3) {
        const int P3 = pwconv_stat_units(d->N, d->mid, HW);
        rc = pwconv_fwd_stat(a2, d->w2, d->ldw2, y3, d->N, d->mid, d->Cout, HW, d->dtype, StatOut{(float2*)workspace, P3},
                             stream);
        if (rc) return rc;
        StatView s3 = stat_view(stat_buf, 2, d->mid, d->Cout);
        return bn_fwd_cp(y3, d->residual ? x : nullptr, out, (const float2*)workspace, P3, d->gamma[2], d->beta[2],
                         d->running_mean[2], d->running_var[2], d->bn_momentum[2], d->bn_eps[2], d->bn_training[2], s3.mean,
                         d->N, d->Cout, HW, 0, d->dtype, stream);
    }
    rc = ofasr_pwconv_fwd(a2, d->w2, d->ldw2, y3, d->N, d->mid, d->Cout, HW, d->dtype, stream);
    if (rc) return rc;
    return bn_forward(d, 2, y3, d->residual ? x : nullptr, out, d->Cout, 0, stat_buf, workspace, workspace_bytes,
                      stream);
}

// prezeroed: the caller has cleared the nine gradient buffers on `stream` already (ofasr_mbstack_bwd: one fill for all blocks)
static int mbconv_bwd_impl(const ofasr_mbconv_desc* d, const void* x, const void* act_buf, const float* stat_buf,
                           const void* dout, void* dx, void* tmp_buf, const ofasr_mbconv_grads* g, void* workspace,
                           size_t workspace_bytes, void* stream, bool prezeroed) {
    const char* name = "ofasr_mbconv_bwd";
    int rc = check_desc(name, d);
    if (rc) return rc;
    OFASR_REQUIRE(x && act_buf && stat_buf && dout && dx && tmp_buf && g, OFASR_ERR_INVALID_ARG, "%s: null buffer", name);
    OFASR_REQUIRE(g->dw1 && g->dw2 && g->dwdw_max, OFASR_ERR_INVALID_ARG, "%s: null weight gradient", name);
    for (int i = 0; i < 3; ++i)
        OFASR_REQUIRE(g->dgamma[i] && g->dbeta[i], OFASR_ERR_INVALID_ARG, "%s: null BN gradient %d", name, i);
    const MbSizes s = mb_sizes(d);
    OFASR_REQUIRE(workspace && workspace_bytes >= s.total, OFASR_ERR_WORKSPACE, "%s: workspace %zu B < required %zu B",
                  name, workspace_bytes, s.total);
    const int64_t HW = d->H * d->W;
    hipStream_t st = as_stream(stream);
    const char* a = (const char*)act_buf;
    const ActLayout lay = act_layout(d, s);
    OFASR_REQUIRE(!lay.fused || (reinterpret_cast<uintptr_t>(act_buf) & 15) == 0, OFASR_ERR_INVALID_ARG,
                  "%s: act_buf must be 16-byte aligned", name);
    const void* y1 = a + lay.y1 * s.es;
    const void* a1 = a + lay.a1 * s.es;
    const void* y2 = a + lay.y2 * s.es;
    const void* a2 = a + lay.a2 * s.es;
    const void* y3 = a + lay.y3 * s.es;
    char* t = (char*)tmp_buf;
    void* tA = t;                                    // mid: da2 -> dy2 (in place)
    void* tB = t + s.mid_elems * s.es;               // mid: da1 -> dy1 (in place)
    void* t3 = t + 2 * s.mid_elems * s.es;           // Cout: dy3
    float* sb = const_cast<float*>(stat_buf);
    const float* f = stat_buf + 8 * d->mid + 4 * d->Cout;
    const int kmax = d->ks[0];

    // dense parameter gradients: zero everything, the kernels fill the active slices.  The host mirror hands out
    // the nine buffers as slices of one allocation: then a single fill covers them (one launch instead of ~18).
    struct Span { char* p; size_t n; };
    const Span sp[9] = {{(char*)g->dw1, (size_t)d->Cmid_max * d->ldw1 * sizeof(float)},
                        {(char*)g->dw2, (size_t)d->Cout_max * d->ldw2 * sizeof(float)},
                        {(char*)g->dwdw_max, (size_t)d->Cmid_max * kmax * kmax * sizeof(float)},
                        {(char*)g->dgamma[0], (size_t)d->Cmid_max * sizeof(float)},
                        {(char*)g->dbeta[0], (size_t)d->Cmid_max * sizeof(float)},
                        {(char*)g->dgamma[1], (size_t)d->Cmid_max * sizeof(float)},
                        {(char*)g->dbeta[1], (size_t)d->Cmid_max * sizeof(float)},
                        {(char*)g->dgamma[2], (size_t)d->Cout_max * sizeof(float)},
                        {(char*)g->dbeta[2], (size_t)d->Cout_max * sizeof(float)}};
    char *lo = sp[0].p, *hi = sp[0].p + sp[0].n;
    size_t sum = 0;
    for (const Span& q : sp) {
        lo = q.p < lo ? q.p : lo;
        hi = q.p + q.n > hi ? q.p + q.n : hi;
        sum += q.n;
    }
    // unjoined side work of earlier calls that still reads the scratch this call is about to overwrite
    SideStream& ss = side_stream();
    const char* t_lo = (const char*)tmp_buf;
    const char* t_hi = t_lo + (3 * s.mid_elems + (size_t)d->N * d->Cout * HW) * s.es;
    const char* w_lo = (const char*)workspace;
    const char* w_hi = w_lo + workspace_bytes;
    bool defer = false;
    {
        std::lock_guard<std::mutex> lk(ss.mu);
        defer = ss.enabled && ss.defer;
        auto hits = [](const char* a0, const char* a1, const char* b0, const char* b1) { return a0 < b1 && b0 < a1; };
        for (const PendingSide& p : ss.pending) {
            if (hits(t_lo, t_hi, p.t0, p.t1) || hits(w_lo, w_hi, p.w0, p.w1) || hits(t_lo, t_hi, p.w0, p.w1) ||
                hits(w_lo, w_hi, p.t0, p.t1)) {
                hipError_t ew = hipStreamWaitEvent(st, p.done, 0);
                OFASR_REQUIRE(ew == hipSuccess, OFASR_ERR_LAUNCH, "%s: wait for deferred side work failed: %s", name,
                              hipGetErrorString(ew));
            }
        }
        if (defer && ss.pending.size() >= MAX_PENDING) {   // bounded bookkeeping: fold everything into one join
            rc = join_side_locked(ss, st, name);
            if (rc) return rc;
        }
    }
    // With the side stream the fill leaves the input-gradient chain: the three weight-gradient buffers are written by
    // side-stream kernels only, so they are cleared there (after fork(0) below); the six BN-gradient vectors are written
    // in full by the chain's BN kernels unless the BN is sliced (then their tails are cleared here).
    auto fill = [&](int i0, int i1, hipStream_t stream_) -> hipError_t {
        char *l = sp[i0].p, *h = sp[i0].p + sp[i0].n;
        size_t tot = 0;
        for (int i = i0; i < i1; ++i) {
            l = sp[i].p < l ? sp[i].p : l;
            h = sp[i].p + sp[i].n > h ? sp[i].p + sp[i].n : h;
            tot += sp[i].n;
        }
        if ((size_t)(h - l) == tot) return hipMemsetAsync(l, 0, tot, stream_);   // adjacent spans: one fill
        hipError_t er = hipSuccess;
        for (int i = i0; i < i1 && er == hipSuccess; ++i) er = hipMemsetAsync(sp[i].p, 0, sp[i].n, stream_);
        return er;
    };
    const bool split_fill = ss.enabled;
    const bool bn_sliced = d->mid < d->Cmid_max || d->Cout < d->Cout_max;
    hipError_t e = hipSuccess;
    if (prezeroed) {
        // nothing to clear
    } else if (!split_fill) {
        if ((size_t)(hi - lo) == sum) {   // disjoint spans (validated above) tiling [lo, hi) exactly
            e = hipMemsetAsync(lo, 0, sum, st);
        } else {
            for (int i = 0; i < 9 && e == hipSuccess; ++i) e = hipMemsetAsync(sp[i].p, 0, sp[i].n, st);
        }
    } else if (bn_sliced) {
        e = fill(3, 9, st);
    }
    OFASR_REQUIRE(e == hipSuccess, OFASR_ERR_LAUNCH, "%s: memset failed: %s", name, hipGetErrorString(e));

    // BN3 (+shortcut, no activation): dy3; the shortcut's gradient is dout itself
    StatView s3 = stat_view(sb, 2, d->mid, d->Cout);
    rc = ofasr_bn_act_bwd(dout, y3, nullptr, t3, nullptr, s3.scale, s3.shift, s3.mean, s3.invstd, g->dgamma[2],
                          g->dbeta[2], d->N, d->Cout, HW, 0, d->bn_training[2], d->dtype, workspace, s.scratch, stream);
    if (rc) return rc;
    // From here the weight gradients go to the side stream: fork after the tensor they read is final, join at the end.
    const bool par = ss.enabled;
    void* sst = par ? (void*)ss.s : stream;   // stream of the weight-gradient kernels
    char* side_ws = (char*)workspace + s.scratch;
    float* dfp = (float*)(side_ws + s.side);
    char* kt_ws = side_ws + s.side + s.df_bytes;
    auto fork = [&](int i) -> int {
        if (!par) return OFASR_OK;
        hipError_t e1 = hipEventRecord(ss.fork[i], st);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(ss.s, ss.fork[i], 0);
        OFASR_REQUIRE(e1 == hipSuccess, OFASR_ERR_LAUNCH, "%s: stream fork failed: %s", name, hipGetErrorString(e1));
        return OFASR_OK;
    };
    const bool fused = lay.fused;
    if (fused)
        OFASR_REQUIRE((reinterpret_cast<uintptr_t>(tmp_buf) & 15) == 0, OFASR_ERR_UNSUPPORTED,
                      "%s: tmp_buf must be 16-byte aligned (the forward pass did not materialise the activations)", name);
    // project 1x1: weight gradient (side) beside input gradient + BN2 backward (main)
    rc = fork(0);   // dy3 (t3) is final
    if (rc) return rc;
    if (split_fill && !prezeroed) {
        const hipError_t ef = fill(0, 3, (hipStream_t)sst);
        OFASR_REQUIRE(ef == hipSuccess, OFASR_ERR_LAUNCH, "%s: memset failed: %s", name, hipGetErrorString(ef));
    }
    if (fused)
        rc = pwconv_wgrad_xf(t3, y2, g->dw2, d->ldw2, d->N, d->mid, d->Cout, HW, d->dtype,
                             xf_of(stat_buf, 1, d->mid, d->Cout), side_ws, s.side, sst);
    else
        rc = ofasr_pwconv_wgrad(t3, a2, g->dw2, d->ldw2, d->N, d->mid, d->Cout, HW, d->dtype, side_ws, s.side, sst);
    if (rc) return rc;
    StatView s2 = stat_view(sb, 1, d->mid, d->Cout);
    StatView s1 = stat_view(sb, 0, d->mid, d->Cout);
    // BN1 / BN2 backward without the apply pass (OFASR_MBCONV_BN_BWD_FOLD=0 restores it): the reduction pass leaves
    // (ka, kbi) per channel; the consumer of dy2 / dy1 on this chain -- the depthwise / expand input gradient -- forms the
    // gradient from (da, y) as it reads them (BwdXf) and stores it once for the weight-gradient kernel of the side
    // stream.  Per BN the chain loses a pass ("read da, read y, write dy"), a launch and one tensor of traffic.  (Reading
    // (da, y) in the weight-gradient kernels too, instead of the stored dy, was measured slower: 2122 against 2261 img/s --
    // those kernels bound the side stream, which then bounds the step.)
    const bool bn_fold = bn_fold_enabled();
    float* coef = reinterpret_cast<float*>((char*)workspace + s.coef_off);
    void* tC = t + (2 * s.mid_elems + (size_t)d->N * d->Cout * HW) * s.es;   // mid: dy2, left there by the depthwise dgrad
    // Where the depthwise weight gradient runs on the matrix cores it forms dy2 from (da2, y2) itself (side stream), so the
    // depthwise input gradient stores no dy2: tA keeps da2 for that kernel and dy1 goes to tC instead of over the dead da2.
    const bool wg_bx = fused && bn_fold_enabled() &&
                       dwconv_wgrad_bx_supported(tA, y1, y2, d->N, d->mid, d->H, d->W, d->K, d->dtype);
    BwdXf bx2{y2, s2.mean, s2.scale, s2.shift, coef, coef + d->mid, wg_bx ? nullptr : tC};
    // Optionally the expand weight gradient forms dy1 from (da1, y1) as well (pw_wgrad_direct_kernel<T, 3>, side stream), and
    // the expand input gradient -- the longest kernel of the chain -- stores no dy1 (50 MB per block at the north-star
    // shape).  OFF by default (OFASR_MBCONV_WG1_BX=1 enables it): the chain's kernel gets 11 us shorter (54 -> 43 us) but
    // the weight-gradient kernel 39 us longer (28 -> 67 us: ten vector instructions per element on twelve fragments per
    // quad make it issue-bound), and with the host out of the way (round 3) the step is bound by the GPU's total work:
    // 2445 against 2514 images/s, two A/B pairs on one box.  (While the step was host-bound the same switch gained 1.2 %.)
    static const bool wg1_env = [] { const char* e = getenv("OFASR_MBCONV_WG1_BX"); return e && e[0] == '1'; }();
    const bool wg1_bx = fused && bn_fold && wg1_env &&
                        pwconv_wgrad_bx_supported(tB, y1, x, d->Cin, d->mid, HW, d->dtype);
    BwdXf bx1{y1, s1.mean, s1.scale, s1.shift, coef + 2 * d->mid, coef + 3 * d->mid,
              wg1_bx ? nullptr : (wg_bx ? tC : tA)};
    // The consumers fold the reduction pass's partial slabs themselves (BwdXf::fold_*): no coefficient launch between a
    // reduction and its consumer on the chain (2 x 5.7 us of pure latency per block); the depthwise input gradient also
    // publishes (ka2, kbi2) for the side stream's weight gradient.  OFASR_MBCONV_BN_COEF_FOLD=0: the coefficient kernel.
    static const bool coef_fold = [] { const char* e = getenv("OFASR_MBCONV_BN_COEF_FOLD"); return !(e && e[0] == '0'); }();
    auto arm_fold = [&](BwdXf& bx, int P, const StatView& sv, int which) {
        bx.fold_partial = (const double*)workspace;
        bx.fold_P = P;
        bx.fold_C = (int)d->mid;
        bx.fold_training = d->bn_training[which];
        bx.fold_M = (double)d->N * (double)HW;
        bx.fold_invstd = sv.invstd;
        bx.fold_dgamma = g->dgamma[which];
        bx.fold_dbeta = g->dbeta[which];
        bx.coef_ka = const_cast<float*>(bx.ka);
        bx.coef_kbi = const_cast<float*>(bx.kbi);
    };
    const bool bxp = fused && bn_fold && dwconv_xf_supported(tA, tB, d->H, d->W, d->K, d->dtype) &&
                     (reinterpret_cast<uintptr_t>(tC) & 15) == 0 &&
                     pwconv_dgrad_bx_supported(tB, y1, dx, d->residual ? dout : nullptr, d->w1, d->ldw1, d->Cin, d->mid, HW,
                                               d->dtype);
    // ... and, optionally, without the reduction pass where the producer of da can take the sums itself (BwdStatOut): the
    // project input gradient for BN2.  OFF by default (OFASR_MBCONV_BN_BWD_STAT=1 / ofasr_debug_mbconv_bn_bwd_stat(1)
    // enable it): alone the pair costs 38.9 + 9.1 us against 25.0 + 24.3 + 5.5 us (rocprofv3, N=16 64x64 mid 384), but
    // the training step is 1 % SLOWER with it (7.01 against 6.94 ms, three A/B pairs on one box) -- the longer
    // store-bound kernel shares HBM with the side stream's weight-gradient kernels for longer.
    const bool bn_stat = g_bn_bwd_stat.load(std::memory_order_relaxed) != 0;
    const int P2b = pwconv_stat_units(d->N, d->Cout, HW);
    const bool st2 = bxp && bn_stat && pwconv_dgrad_bstat_supported(t3, y2, tA, d->w2, d->ldw2, d->mid, d->Cout, HW, d->dtype) &&
                     (size_t)P2b * (size_t)d->mid * sizeof(float2) <= s.stat_a;
    if (st2) {
        rc = pwconv_dgrad_bstat(t3, d->w2, d->ldw2, tA, d->N, d->mid, d->Cout, HW, d->dtype,
                                BwdStatOut{y2, s2.mean, s2.scale, s2.shift, (float2*)workspace, P2b}, stream);
        if (rc) return rc;
        rc = bn_bwd_coef_cp((const float2*)workspace, P2b, d->mid, (double)d->N * (double)HW, d->bn_training[1], s2.scale,
                            s2.invstd, g->dgamma[1], g->dbeta[1], coef, coef + d->mid, stream);
    } else {
        rc = ofasr_pwconv_dgrad(t3, d->w2, d->ldw2, tA, d->N, d->mid, d->Cout, HW, d->dtype, stream);
    }
    if (rc) return rc;
    if (bxp) {
        if (!st2 && coef_fold) {
            int P2 = 0;
            rc = bn_bwd_reduce_only(tA, y2, s2.scale, s2.shift, s2.mean, s2.invstd, d->N, d->mid, HW, 1, d->dtype, workspace,
                                    s.scratch, &P2, stream);
            if (rc) return rc;
            arm_fold(bx2, P2, s2, 1);
        } else if (!st2) {
            rc = bn_bwd_reduce_coef(tA, y2, s2.scale, s2.shift, s2.mean, s2.invstd, g->dgamma[1], g->dbeta[1], coef,
                                    coef + d->mid, d->N, d->mid, HW, 1, d->bn_training[1], d->dtype, workspace, s.scratch,
                                    stream);
            if (rc) return rc;
        }
        rc = dwconv_dgrad_bx(tA, f, tB, d->N, d->mid, d->H, d->W, d->K, d->dtype, bx2, stream);   // also leaves dy2 in tC
        if (rc) return rc;
        rc = fork(1);   // dy2 (tC), or what the weight gradient forms it from (da2 in tA, the BN2 coefficients), is final
        if (rc) return rc;
        if (wg_bx)
            rc = dwconv_wgrad_xf_bx(tA, y1, dfp, d->N, d->mid, d->H, d->W, d->K, d->dtype, xf_of(stat_buf, 0, d->mid, d->Cout),
                                    BwdXf{y2, s2.mean, s2.scale, s2.shift, coef, coef + d->mid, nullptr}, sst);
        else
            rc = dwconv_wgrad_xf(tC, y1, dfp, d->N, d->mid, d->H, d->W, d->K, d->dtype, xf_of(stat_buf, 0, d->mid, d->Cout),
                                 side_ws, s.side, sst);
        if (rc) return rc;
        rc = (t_skip_kt & 2) ? OFASR_OK
                             : ofasr_ktransform_bwd(d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, dfp, g->dwdw_max,
                                                    g->dmats, d->mid, kt_ws, s.ws_kt + 256, sst);
        if (rc) return rc;
        if (coef_fold) {
            int P1 = 0;
            rc = bn_bwd_reduce_only(tB, y1, s1.scale, s1.shift, s1.mean, s1.invstd, d->N, d->mid, HW, 1, d->dtype, workspace,
                                    s.scratch, &P1, stream);
            if (rc) return rc;
            arm_fold(bx1, P1, s1, 0);
        } else {
            rc = bn_bwd_reduce_coef(tB, y1, s1.scale, s1.shift, s1.mean, s1.invstd, g->dgamma[0], g->dbeta[0], coef + 2 * d->mid,
                                    coef + 3 * d->mid, d->N, d->mid, HW, 1, d->bn_training[0], d->dtype, workspace, s.scratch,
                                    stream);
        }
        if (rc) return rc;
        // expand input gradient (+ the shortcut's dout); leaves dy1 in tA (da2 there is dead: its one reader ran above)
        rc = pwconv_dgrad_add_bx(tB, d->w1, d->ldw1, dx, d->residual ? dout : nullptr, d->N, d->Cin, d->mid, HW, d->dtype, bx1,
                                 stream);
        if (rc) return rc;
        rc = fork(2);   // dy1 is final
        if (rc) return rc;
        if (wg1_bx)   // da1 (tB) and the coefficients the expand input gradient published are final
            rc = pwconv_wgrad_bx(tB, x, g->dw1, d->ldw1, d->N, d->Cin, d->mid, HW, d->dtype,
                                 BwdXf{y1, s1.mean, s1.scale, s1.shift, coef + 2 * d->mid, coef + 3 * d->mid, nullptr}, side_ws,
                                 s.side, sst);
        else
            rc = ofasr_pwconv_wgrad(wg_bx ? tC : tA, x, g->dw1, d->ldw1, d->N, d->Cin, d->mid, HW, d->dtype, side_ws, s.side, sst);
        if (rc) return rc;
    } else {
    // BN2 + ReLU6 (in place: da2 -> dy2)
    rc = ofasr_bn_act_bwd(tA, y2, nullptr, tA, nullptr, s2.scale, s2.shift, s2.mean, s2.invstd, g->dgamma[1],
                          g->dbeta[1], d->N, d->mid, HW, 1, d->bn_training[1], d->dtype, workspace, s.scratch, stream);
    if (rc) return rc;
    // depthwise: filter gradient and its chain back to the max-size weight / matrices (side) beside the input
    // gradient + BN1 backward (main)
    rc = fork(1);   // dy2 (tA) is final
    if (rc) return rc;
    if (fused)
        rc = dwconv_wgrad_xf(tA, y1, dfp, d->N, d->mid, d->H, d->W, d->K, d->dtype, xf_of(stat_buf, 0, d->mid, d->Cout),
                             side_ws, s.side, sst);
    else
        rc = ofasr_dwconv_wgrad(tA, a1, dfp, d->N, d->mid, d->H, d->W, d->K, d->dtype, side_ws, s.side, sst);
    if (rc) return rc;
    rc = (t_skip_kt & 2) ? OFASR_OK
                         : ofasr_ktransform_bwd(d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, dfp, g->dwdw_max,
                                                g->dmats, d->mid, kt_ws, s.ws_kt + 256, sst);
    if (rc) return rc;
    rc = ofasr_dwconv_dgrad(tA, f, tB, d->N, d->mid, d->H, d->W, d->K, d->dtype, stream);
    if (rc) return rc;
    // BN1 + ReLU6 (in place: da1 -> dy1)
    rc = ofasr_bn_act_bwd(tB, y1, nullptr, tB, nullptr, s1.scale, s1.shift, s1.mean, s1.invstd, g->dgamma[0],
                          g->dbeta[0], d->N, d->mid, HW, 1, d->bn_training[0], d->dtype, workspace, s.scratch, stream);
    if (rc) return rc;
    // expand 1x1: weight gradient (side) beside the input gradient (main); x also feeds the identity shortcut, whose
    // gradient (dout) is added in the dgrad epilogue
    rc = fork(2);   // dy1 (tB) is final
    if (rc) return rc;
    rc = ofasr_pwconv_wgrad(tB, x, g->dw1, d->ldw1, d->N, d->Cin, d->mid, HW, d->dtype, side_ws, s.side, sst);
    if (rc) return rc;
    if (d->residual)
        rc = pwconv_dgrad_add(tB, d->w1, d->ldw1, dx, dout, d->N, d->Cin, d->mid, HW, d->dtype, stream);
    else
        rc = ofasr_pwconv_dgrad(tB, d->w1, d->ldw1, dx, d->N, d->Cin, d->mid, HW, d->dtype, stream);
    if (rc) return rc;
    }
    if (par) {
        std::lock_guard<std::mutex> lk(ss.mu);
        if (defer) {   // remember what the side kernels still use; ofasr_mbconv_join orders them before the caller
            const size_t k = ss.pending.size();
            if (ss.pool.size() <= k) {
                hipEvent_t ev = nullptr;
                const hipError_t e3 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e3 != hipSuccess) {   // no event: fall back to the immediate join
                    (void)hipGetLastError();
                    return join_side_locked(ss, st, name);
                }
                ss.pool.push_back(ev);
            }
            const hipError_t e3 = hipEventRecord(ss.pool[k], ss.s);
            OFASR_REQUIRE(e3 == hipSuccess, OFASR_ERR_LAUNCH, "%s: event record failed: %s", name, hipGetErrorString(e3));
            ss.pending.push_back(PendingSide{t_lo, t_hi, w_lo, w_hi, ss.pool[k]});
        } else {       // join: everything this call enqueued is ordered before whatever the caller enqueues next
            rc = join_side_locked(ss, st, name);
        }
    }
    return rc;
}

// ---- the MB stack: every active block of the network in ONE host call per direction --------------------------------
// Block i reads the output of block i - 1 (the last N*Cout*HW elements of its act_buf); in the backward its dout is
// the dx of block i + 1.  Nothing but a loop over ofasr_mbconv_fwd / _bwd: what it removes is the per-block host path
// above the C ABI (autograd node, descriptor marshalling, three allocations, the foreign call itself).
static const void* stack_out_of(const ofasr_mbstack_item& it) {
    const ofasr_mbconv_desc* d = it.desc;
    const size_t es = d->dtype == OFASR_F32 ? 4 : 2;
    const size_t out_elems = (size_t)(d->N * d->Cout * d->H * d->W);
    return (const char*)it.act_buf + (ofasr_mbconv_act_elems(d) - out_elems) * es;
}

static bool stack_kt_batch_enabled() {
    static const bool on = [] { const char* e = getenv("OFASR_MBSTACK_KT_BATCH"); return !(e && e[0] == '0'); }();
    return on;
}

OFASR_EXPORT int ofasr_mbstack_fwd(const ofasr_mbstack_item* items, int n, const void* x, void* stream) {
    OFASR_REQUIRE(items && n > 0 && x, OFASR_ERR_INVALID_ARG, "ofasr_mbstack_fwd: null / empty stack");
    for (int i = 0; i < n; ++i) {
        OFASR_REQUIRE(items[i].desc != nullptr && items[i].stat_buf != nullptr, OFASR_ERR_INVALID_ARG,
                      "ofasr_mbstack_fwd: block %d has no descriptor / statistics buffer", i);
        if (i > 0) {
            const ofasr_mbconv_desc *a = items[i - 1].desc, *b = items[i].desc;
            OFASR_REQUIRE(a->N == b->N && a->Cout == b->Cin && a->H == b->H && a->W == b->W && a->dtype == b->dtype,
                          OFASR_ERR_INVALID_ARG, "ofasr_mbstack_fwd: block %d does not take block %d's output", i, i - 1);
        }
    }
    // the active depthwise filters of all blocks (centre crop + transform chain) in ONE launch, ahead of the blocks
    const bool batch = stack_kt_batch_enabled() && n <= 16;
    if (batch) {
        KtJob jobs[16];
        for (int i = 0; i < n; ++i) {
            const ofasr_mbconv_desc* d = items[i].desc;
            int rc = check_desc("ofasr_mbstack_fwd", d);
            if (rc) return rc;
            jobs[i] = KtJob{d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, d->mid,
                            items[i].stat_buf + 8 * d->mid + 4 * d->Cout, nullptr, nullptr, nullptr, nullptr, 0};
        }
        int rc = ktransform_fwd_batch(jobs, n, stream);
        if (rc) return rc;
    }
    const void* in = x;
    const int was = t_skip_kt;
    if (batch) t_skip_kt |= 1;
    int rc = OFASR_OK;
    for (int i = 0; i < n && rc == OFASR_OK; ++i) {
        const ofasr_mbstack_item& it = items[i];
        rc = ofasr_mbconv_fwd(it.desc, in, it.act_buf, it.stat_buf, it.workspace, it.workspace_bytes, stream);
        in = stack_out_of(it);
    }
    t_skip_kt = was;
    return rc;
}

OFASR_EXPORT int ofasr_mbstack_bwd(const ofasr_mbstack_item* items, int n, const void* x, const void* dout, void* stream) {
    OFASR_REQUIRE(items && n > 0 && x && dout, OFASR_ERR_INVALID_ARG, "ofasr_mbstack_bwd: null / empty stack");
    // The dense parameter gradients of ALL blocks: when the caller handed them out as slices of one allocation (the host
    // mirror does) they are cleared with ONE fill on the caller's stream, ahead of every kernel of either stream (the side
    // stream's kernels wait for fork events recorded later on this stream) -- instead of two fills per block.
    bool prezeroed = false;
    {
        const char *lo = nullptr, *hi = nullptr;
        size_t sum = 0;
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const ofasr_mbconv_desc* d = items[i].desc;
            const ofasr_mbconv_grads* g = items[i].grads;
            if (!d || !g || !g->dw1 || !g->dw2 || !g->dwdw_max) { ok = false; break; }
            const int kmax = d->ks[0];
            const struct { const void* p; size_t nbytes; } sp[9] = {
                {g->dw1, (size_t)d->Cmid_max * d->ldw1 * sizeof(float)}, {g->dw2, (size_t)d->Cout_max * d->ldw2 * sizeof(float)},
                {g->dwdw_max, (size_t)d->Cmid_max * kmax * kmax * sizeof(float)},
                {g->dgamma[0], (size_t)d->Cmid_max * sizeof(float)}, {g->dbeta[0], (size_t)d->Cmid_max * sizeof(float)},
                {g->dgamma[1], (size_t)d->Cmid_max * sizeof(float)}, {g->dbeta[1], (size_t)d->Cmid_max * sizeof(float)},
                {g->dgamma[2], (size_t)d->Cout_max * sizeof(float)}, {g->dbeta[2], (size_t)d->Cout_max * sizeof(float)}};
            for (const auto& q : sp) {
                if (!q.p) { ok = false; break; }
                const char* p = (const char*)q.p;
                lo = (!lo || p < lo) ? p : lo;
                hi = (!hi || p + q.nbytes > hi) ? p + q.nbytes : hi;
                sum += q.nbytes;
            }
        }
        // one fill when the span is at most the transform-matrix gradients (which the kernels write in full) larger
        if (ok && lo && (size_t)(hi - lo) >= sum && (size_t)(hi - lo) <= sum + (size_t)n * 3 * 1024 * sizeof(float)) {
            const hipError_t e = hipMemsetAsync(const_cast<char*>(lo), 0, (size_t)(hi - lo), as_stream(stream));
            OFASR_REQUIRE(e == hipSuccess, OFASR_ERR_LAUNCH, "ofasr_mbstack_bwd: memset failed: %s", hipGetErrorString(e));
            prezeroed = true;
        }
    }
    const void* g = dout;
    const bool batch = stack_kt_batch_enabled() && n <= 16;
    const int was = t_skip_kt;
    if (batch) t_skip_kt |= 2;
    int rc = OFASR_OK;
    for (int i = n - 1; i >= 0 && rc == OFASR_OK; --i) {
        const ofasr_mbstack_item& it = items[i];
        if (!(it.desc && it.dx && it.tmp_buf && it.grads)) {
            set_error("ofasr_mbstack_bwd: block %d lacks a backward buffer", i);
            rc = OFASR_ERR_INVALID_ARG;
            break;
        }
        const void* in = i > 0 ? stack_out_of(items[i - 1]) : x;
        rc = mbconv_bwd_impl(it.desc, in, it.act_buf, it.stat_buf, g, it.dx, it.tmp_buf, it.grads, it.workspace,
                             it.workspace_bytes, stream, prezeroed);
        g = it.dx;
    }
    t_skip_kt = was;
    if (rc || !batch) return rc;
    // the filter gradients' way back through the transform chains (dense weight gradient + matrix gradients) of all blocks:
    // two launches on the stream the depthwise weight gradients ran on, after the last of them
    SideStream& ss = side_stream();
    void* sst = ss.enabled ? (void*)ss.s : stream;
    KtJob jobs[16];
    for (int i = 0; i < n; ++i) {
        const ofasr_mbconv_desc* d = items[i].desc;
        const MbSizes sz = mb_sizes(d);
        char* side_ws = (char*)items[i].workspace + sz.scratch;
        jobs[i] = KtJob{d->wdw_max, d->ks, d->chain_len - 1, d->mats, d->transform, d->mid, nullptr,
                        (const float*)(side_ws + sz.side), items[i].grads->dwdw_max, items[i].grads->dmats,
                        side_ws + sz.side + sz.df_bytes, sz.ws_kt + 256};
    }
    rc = ktransform_bwd_batch(jobs, n, sst);
    if (rc || !ss.enabled) return rc;
    {
        std::lock_guard<std::mutex> lk(ss.mu);
        if (ss.defer) {   // one more entry behind the blocks': ofasr_mbconv_join orders this launch before the caller as well
            const size_t k = ss.pending.size();
            if (ss.pool.size() <= k) {
                hipEvent_t ev = nullptr;
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    return join_side_locked(ss, as_stream(stream), "ofasr_mbstack_bwd");
                }
                ss.pool.push_back(ev);
            }
            const hipError_t e3 = hipEventRecord(ss.pool[k], ss.s);
            OFASR_REQUIRE(e3 == hipSuccess, OFASR_ERR_LAUNCH, "ofasr_mbstack_bwd: event record failed: %s", hipGetErrorString(e3));
            // (it reads every block's filter gradient and chain scratch: the span of the blocks' workspaces)
            const char *wlo = nullptr, *whi = nullptr;
            for (int i = 0; i < n; ++i) {
                const char* w0 = (const char*)items[i].workspace;
                wlo = (!wlo || w0 < wlo) ? w0 : wlo;
                whi = (!whi || w0 + items[i].workspace_bytes > whi) ? w0 + items[i].workspace_bytes : whi;
            }
            ss.pending.push_back(PendingSide{nullptr, nullptr, wlo, whi, ss.pool[k]});
        } else {
            rc = join_side_locked(ss, as_stream(stream), "ofasr_mbstack_bwd");
        }
    }
    return rc;
}

OFASR_EXPORT int ofasr_mbconv_bwd(const ofasr_mbconv_desc* d, const void* x, const void* act_buf, const float* stat_buf,
                                  const void* dout, void* dx, void* tmp_buf, const ofasr_mbconv_grads* g,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return mbconv_bwd_impl(d, x, act_buf, stat_buf, dout, dx, tmp_buf, g, workspace, workspace_bytes, stream, false);
}

OFASR_EXPORT int ofasr_debug_mbconv_bn_bwd_stat(int enable) {
    return g_bn_bwd_stat.exchange(enable ? 1 : 0, std::memory_order_relaxed);
}

OFASR_EXPORT int ofasr_mbconv_defer_join(int enable) {
    SideStream& ss = side_stream();
    std::lock_guard<std::mutex> lk(ss.mu);
    const int was = ss.defer ? 1 : 0;
    if (!enable && ss.defer && !ss.pending.empty()) {   // leaving the mode with work in flight: drain it (blocking)
        (void)hipStreamSynchronize(ss.s);
        ss.pending.clear();
    }
    ss.defer = ss.enabled && enable != 0;
    return was;
}

// the side stream itself, for callers that put further independent work (e.g. the static convs' weight gradients)
// beside the composite calls; NULL when OFASR_MBCONV_SIDE_STREAM=0.  Ordering such work is the caller's business.
OFASR_EXPORT void* ofasr_side_stream(void) {
    SideStream& ss = side_stream();
    return ss.enabled ? (void*)ss.s : nullptr;
}

OFASR_EXPORT int ofasr_mbconv_join(void* stream) {
    SideStream& ss = side_stream();
    std::lock_guard<std::mutex> lk(ss.mu);
    if (ss.pending.empty()) return OFASR_OK;
    return join_side_locked(ss, as_stream(stream), "ofasr_mbconv_join");
}
