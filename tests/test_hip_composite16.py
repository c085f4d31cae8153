"""GPU parity of the composite MB-block call with 16-bit activations -- the path bench.py times -- against the CPU
oracle, at BASELINE shapes ([N,64,64,64] and [N,64,32,32], mid in {192,256,384}, k in {3,5,7}, train- and eval-mode
BN, identity shortcut).  `-m gpu`.

`ofasr_mbconv_fwd/_bwd` (include/ofasr.h) are reached through ops.FusedMBConvFn exactly as the trainer reaches them.
Every stage of the call is checked on its own: the oracle stage (oracle/composite16.py -- the C oracle's operators,
double accumulation, BN in double) is fed the 16-bit tensors the GPU stage READ (taken from the call's own buffers) and
its result, rounded once, is compared with what the GPU stage WROTE, element by element at the 16-bit tolerances of
DESIGN.md section 4 (rtol 1e-2 bf16 / 2e-3 f16).  Reference call sites restated by the oracle:
dynamic_layers.py:70-84, dynamic_op.py:46-84,104-112,148-167, proxyless_nets.py:44-51.  The launch counters of the
library (include/ofasr.h, Diagnostics) assert that the kernel variants of the timed path served the call."""
import os

import numpy as np
import pytest
import torch

from conftest import amd, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

RT = {torch.bfloat16: 1e-2, torch.float16: 2e-3}
TNAME = {torch.bfloat16: "ofasr::bf16_t", torch.float16: "ofasr::f16_t"}


def H(t):
    return t.detach().float().cpu().numpy()


def close16(got, ref, dtype, what, safe=None, frac_ok=1e-4):
    """element-wise |got - ref| <= rtol*|ref| + rtol*rms(ref); at most `frac_ok` of the elements may sit between 1x and
    3x that bound (a double rounding that lands on the other side of a 16-bit tie), none beyond."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rt = RT[dtype]
    tol = rt * np.abs(ref) + rt * float(np.sqrt(np.mean(ref * ref)) + 1e-30)
    err = np.abs(got - ref)
    if safe is not None:
        err = np.where(safe, err, 0.0)
    bad = err > tol
    assert bad.mean() <= frac_ok, "%s: %.3g of the elements outside the 16-bit tolerance" % (what, bad.mean())
    assert not np.any(err > 3 * tol), "%s: max |err| %.4g at tol %.4g" % (
        what, err.max(), tol.flat[int(np.argmax(err - 3 * tol))])


def close32(got, ref, what, rt=2e-3):
    """fp32 outputs (parameter gradients, statistics): sums of products of 16-bit operands accumulated in fp32"""
    ref = np.asarray(ref, np.float64)
    assert_close(got, ref, rt, rt * float(np.sqrt(np.mean(ref * ref)) + 1e-30), what)


def _make_block(seed):
    dop = amd("elastic_nn.modules.dynamic_op")
    dl = amd("elastic_nn.modules.dynamic_layers")
    blk = amd("imagenet_codebase.networks")
    layers = amd("layers")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    g = torch.Generator().manual_seed(seed)
    layer = dl.DynamicMBConvLayer([64], [64], [3, 5, 7], [3, 4, 6])
    block = blk.MobileInvertedResidualBlock(layer, layers.IdentityLayer([64], [64]))
    with torch.no_grad():
        for name, p in block.named_parameters():
            if name.endswith("_matrix"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[0] * p.shape[2] * p.shape[3])) ** 0.5 * 2.0)
            elif name.endswith("weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            else:
                p.copy_(0.6 * torch.rand(p.shape, generator=g) - 0.1)
        for name, b in block.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.2 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return block, layer


CASES = [
    # (N, H, W, expand, K, train, dtype)
    (2, 64, 64, 6, 7, True, torch.bfloat16), (2, 64, 64, 6, 5, True, torch.bfloat16),
    (2, 64, 64, 6, 3, True, torch.bfloat16), (2, 64, 64, 4, 7, True, torch.bfloat16),
    (2, 64, 64, 4, 5, True, torch.bfloat16), (2, 64, 64, 4, 3, True, torch.bfloat16),
    (2, 64, 64, 3, 7, True, torch.bfloat16), (2, 64, 64, 3, 5, True, torch.bfloat16),
    (2, 64, 64, 3, 3, True, torch.bfloat16),
    (2, 64, 64, 6, 7, False, torch.bfloat16), (2, 64, 64, 4, 5, False, torch.bfloat16),
    (2, 64, 64, 3, 3, False, torch.bfloat16),
    (3, 32, 32, 6, 7, True, torch.bfloat16), (3, 32, 32, 4, 5, True, torch.bfloat16),
    (3, 32, 32, 3, 3, True, torch.bfloat16), (3, 32, 32, 6, 5, False, torch.bfloat16),
    (5, 64, 64, 6, 7, True, torch.bfloat16),          # N not a multiple of the images-per-wave of the MFMA depthwise
    (2, 64, 64, 6, 7, True, torch.float16), (2, 64, 64, 4, 5, True, torch.float16),
    (2, 64, 64, 3, 3, True, torch.float16), (2, 64, 64, 6, 5, False, torch.float16),
    (3, 32, 32, 6, 7, True, torch.float16),
    (2, 48, 48, 6, 3, True, torch.bfloat16),          # BASELINE config 2's LR size
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_%dx%d_e%d_k%d_%s_%s" % (
    c[0], c[1], c[2], c[3], c[4], "train" if c[5] else "eval", "bf16" if c[6] == torch.bfloat16 else "f16"))
@pytest.mark.parametrize("bstat", [False, True], ids=["", "bn2sums_in_dgrad"])
def test_composite_block_16bit_vs_oracle(ora, case, bstat):
    from oracle import composite16 as c16
    N, Hh, Ww, e, K, train, dtype = case
    ops, C = amd("ops"), amd("_C")
    if bstat and not (train and (Hh * Ww) % 128 == 0 and 64 * e in (256, 384)):
        pytest.skip("the BN2-sums-in-the-project-dgrad variant (off by default) only exists for the slab-walk kernel")
    was_bstat = C.lib().ofasr_debug_mbconv_bn_bwd_stat(1 if bstat else 0)
    block, layer = _make_block(100 * K + e)
    block.to(DEV).train(train)
    layer.active_kernel_size, layer.active_expand_ratio = K, e
    mid = layer.active_middle_channel(64)
    sd0 = {k: v.detach().cpu().clone() for k, v in block.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    x16 = torch.randn((N, 64, Hh, Ww), generator=g).to(dtype)
    do16 = (0.05 * torch.randn((N, 64, Hh, Ww), generator=g)).to(dtype)

    was_tmp, was_defer = ops.SHARED_TMP, ops.deferred_weight_grads(True)
    ops.SHARED_TMP = True
    ops._TMP_CACHE.clear()
    try:
        C.reset_launch_counts()
        xg = x16.to(DEV).requires_grad_(True)
        y = block(xg)
        assert type(y.grad_fn).__name__.startswith("FusedMBConvFn"), "the composite call did not serve the block"
        saved = y.grad_fn.saved_tensors
        act, stat = saved[1], saved[2]
        fwd_table = C.launch_table()
        C.reset_launch_counts()
        y.backward(do16.to(DEV))
        amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
        torch.cuda.synchronize()
        bwd_table = C.launch_table()
        (tmp,) = list(ops._TMP_CACHE.values())
    finally:
        ops.SHARED_TMP = was_tmp
        ops.deferred_weight_grads(was_defer)
        ops._TMP_CACHE.clear()
        C.lib().ofasr_debug_mbconv_bn_bwd_stat(was_bstat)

    # ---- routing: the kernel variants bench.py's timed step runs (profiles/r02_*_steps.txt) served this call
    T = TNAME[dtype]

    def ran(table, *parts):
        return sum(n for name, n in table.items() if all(p in name for p in parts))

    aligned = (Hh * Ww) % 128 == 0
    if train and aligned:
        if mid in (256, 384):
            assert ran(fwd_table, "pw_fanout_slabs_kernel<%s, %d, 1>" % (T, mid // 128)) == 1, fwd_table   # + BN1 stats
            if bstat:   # project dgrad + the BN2-backward sums of what it writes (BwdStatOut): no reduction pass for BN2
                assert ran(bwd_table, "pw_fanout_slabs_kernel<%s, %d, 2>" % (T, mid // 128)) == 1, bwd_table
                assert ran(bwd_table, "bn_bwd_coef_cp_kernel") == 1, bwd_table
            else:
                assert ran(bwd_table, "pw_fanout_slabs_kernel<%s, %d, 0>" % (T, mid // 128)) == 1, bwd_table
                assert ran(bwd_table, "bn_bwd_coef_cp_kernel") == 0, bwd_table
        assert ran(fwd_table, "pw_fanin_pipe_kernel<%s, true, true, false>" % T) == 1, fwd_table      # project: XF + fold
        # backward: BN1 / BN2 have no apply pass -- their consumers read (da, y) through the BN backward (BwdXf variants)
        assert ran(bwd_table, "pw_fanin_pipe_kernel<%s, false, true, true>" % T) == 1, bwd_table   # expand dgrad (+dout)
        assert ran(bwd_table, "pw_wgrad_direct_kernel<%s, 1>" % T) + ran(bwd_table, "pw_wgrad_direct_kernel<%s, 2>" % T) == 1
        # the expand weight gradient: from the dy1 the expand dgrad stored (<T, 0>, default) or, with OFASR_MBCONV_WG1_BX=1,
        # formed from (da1, y1) as it reads them (BwdXf, <T, 3>; the expand dgrad then stores no dy1)
        want3 = 1 if os.environ.get("OFASR_MBCONV_WG1_BX", "0") == "1" else 0
        assert ran(bwd_table, "pw_wgrad_direct_kernel<%s, 3>" % T) == want3 and \
            ran(bwd_table, "pw_wgrad_direct_kernel<%s, 0>" % T) == 1 - want3, bwd_table
        if Ww in (32, 64) and Hh % 16 == 0 and Hh <= 64:   # depthwise weight gradient on the matrix cores, no reduce launch
            assert ran(bwd_table, "dw_wgrad_mfma_kernel<%s, %d, %d, true, true>" % (T, K, Ww // 32)) == 1, bwd_table
            assert ran(bwd_table, "dw_wgrad_vec") == 0, bwd_table
        else:
            assert ran(bwd_table, "dw_wgrad_vec_kernel<%s, %d, true>" % (T, K)) == 1, bwd_table
        nstat = ran(bwd_table, "bn_bwd_coef_cp_kernel")
        # BN3 (64 channels, no activation): the reduction + apply pair; with OFASR_BN_BWD_ONEPASS=1 (opt-in, measured slower in
        # the step) a one-pass backward from registers when a channel fits a workgroup (N * HW <= 65536)
        one3 = ran(bwd_table, "bn_bwd_onepass_kernel")
        assert one3 == (1 if (N * Hh * Ww <= 65536 and os.environ.get("OFASR_BN_BWD_ONEPASS", "0") == "1") else 0), bwd_table
        assert ran(bwd_table, "bn_bwd_reduce_kernel") == 3 - one3 - nstat and ran(bwd_table, "bn_bwd_apply_kernel") == 1 - one3
        # ... and no coefficient launch either: the consumers fold the reduction's partial slabs themselves (BwdXf::fold_*)
        assert ran(bwd_table, "bn_bwd_coef_kernel") == 0
        if K in (5, 7) and Ww in (32, 64):
            assert ran(fwd_table, "dw_mfma_kernel<%s, %d, false, true, true, false>" % (T, K)) == 1, fwd_table
            assert ran(bwd_table, "dw_mfma_kernel<%s, %d, true, false, false, true>" % (T, K)) == 1, bwd_table
        else:
            assert ran(fwd_table, "dw_vec_kernel<%s, %d, false, true, true, false>" % (T, K)) == 1, fwd_table
            assert ran(bwd_table, "dw_vec_kernel<%s, %d, true, false, false, true>" % (T, K)) == 1, bwd_table
    # only BN3 has an apply pass (or its one-pass form): BN1 / BN2 are folded into their consumers
    folded_bwd = ran(bwd_table, "bn_bwd_apply_kernel") + ran(bwd_table, "bn_bwd_onepass_kernel") == 1
    dw_mma = ran(fwd_table, "dw_mfma_kernel") > 0
    assert ran(fwd_table, "bn_stats_kernel") == 0 and ran(fwd_table, "bn_finalize") == 0 or not (train and aligned)

    # ---- the call's own buffers (layout: csrc/mbconv.hip)
    P = N * Hh * Ww
    a = act.float().cpu().numpy()
    y1 = a[0:P * mid].reshape(N, mid, Hh, Ww)
    nmid = 2 if a.size == P * (2 * mid + 128) else 4     # fused path: y1 | y2 | y3 | out (no room for a1 / a2)
    y2 = a[(nmid // 2) * P * mid:(nmid // 2 + 1) * P * mid].reshape(N, mid, Hh, Ww)
    y3 = a[nmid * P * mid:nmid * P * mid + P * 64].reshape(N, 64, Hh, Ww)
    out = a[nmid * P * mid + P * 64:].reshape(N, 64, Hh, Ww)
    assert np.array_equal(out, H(y))
    st = stat.cpu().numpy()
    st1, st2, st3 = st[0:4 * mid].reshape(4, mid), st[4 * mid:8 * mid].reshape(4, mid), st[8 * mid:8 * mid + 256].reshape(4, 64)
    f_gpu = st[8 * mid + 256:].reshape(mid, 1, K, K)
    t = tmp.float().cpu().numpy()
    # folded BN backward: tA = dy1 (over the dead da2), tB = da1, tC = dy2;  apply pass: tA = dy2, tB = dy1 (in place)
    tA = t[0:P * mid].reshape(N, mid, Hh, Ww)
    tB = t[P * mid:2 * P * mid].reshape(N, mid, Hh, Ww)
    dy3 = t[2 * P * mid:2 * P * mid + P * 64].reshape(N, 64, Hh, Ww)
    tC = t[2 * P * mid + P * 64:3 * P * mid + P * 64].reshape(N, mid, Hh, Ww)

    pfx = "mobile_inverted_conv."
    w1 = sd0[pfx + "inverted_bottleneck.conv.conv.weight"].numpy()
    w2 = sd0[pfx + "point_linear.conv.conv.weight"].numpy()
    wdw = sd0[pfx + "depth_conv.conv.conv.weight"].numpy()
    mats = {"7to5": sd0[pfx + "depth_conv.conv.7to5_matrix"].numpy(), "5to3": sd0[pfx + "depth_conv.conv.5to3_matrix"].numpy()}
    bnp = {}
    for i, nm in enumerate(("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.", "point_linear.bn.bn.")):
        bnp[i] = {k: sd0[pfx + nm + k].numpy() for k in ("weight", "bias", "running_mean", "running_var")}
    x = x16.float().numpy()
    dout = do16.float().numpy()
    sd1 = {k: v.detach().cpu() for k, v in block.state_dict().items()}
    grads = {n: (None if p.grad is None else p.grad.detach().cpu().numpy()) for n, p in block.named_parameters()}

    def check_stats(i, ytens, stg, name):
        b = bnp[i]
        mean, invstd = c16.bn_consts(ytens, b["weight"], b["bias"], b["running_mean"], b["running_var"], train)
        Cc = ytens.shape[1]
        assert np.all(np.abs(stg[0] - mean) <= 2e-5 / invstd + 1e-6 * np.abs(mean)), name + " mean"
        assert_close(stg[1], invstd, 2e-4, 0, name + " invstd")
        if train:
            rm, rv = c16.running_update(ytens, b["running_mean"], b["running_var"])
            nm = ("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.", "point_linear.bn.bn.")[i]
            assert_close(sd1[pfx + nm + "running_mean"].numpy(), rm, 1e-4, 1e-5, name + " running_mean")
            assert_close(sd1[pfx + nm + "running_var"].numpy(), rv, 2e-4, 1e-6, name + " running_var")
            assert int(sd1[pfx + nm + "num_batches_tracked"]) == int(sd0[pfx + nm + "num_batches_tracked"]) + 1
            assert np.array_equal(sd1[pfx + nm + "running_mean"].numpy()[Cc:], b["running_mean"][Cc:])   # slice only
        else:
            assert torch.equal(sd1[pfx + ("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.", "point_linear.bn.bn.")[i]
                                   + "running_mean"], sd0[pfx + ("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.",
                                                                 "point_linear.bn.bn.")[i] + "running_mean"])
        return mean, invstd

    # ---- forward, stage by stage
    close16(y1, c16.r16(c16.expand_fwd(x, w1, mid, dtype), dtype), dtype, "y1 (expand)")
    m1, i1 = check_stats(0, y1, st1, "BN1")
    ks_set = [3, 5, 7]
    f_ref = ora.ktransform_fwd(wdw, mid, K, ks_set, mats)
    assert_close(f_gpu, f_ref, 2e-5, 2e-6, "active filter")
    a1, pre1 = c16.bn_apply(y1, m1, i1, bnp[0]["weight"], bnp[0]["bias"], True)
    close16(y2, c16.r16(c16.depthwise_fwd(a1, f_gpu, dtype, dw_mma), dtype), dtype, "y2 (depthwise)")
    m2, i2 = check_stats(1, y2, st2, "BN2")
    a2, pre2 = c16.bn_apply(y2, m2, i2, bnp[1]["weight"], bnp[1]["bias"], True)
    close16(y3, c16.r16(c16.project_fwd(a2, w2, 64, dtype), dtype), dtype, "y3 (project)")
    m3, i3 = check_stats(2, y3, st3, "BN3")
    o_ref, _ = c16.bn_apply(y3, m3, i3, bnp[2]["weight"], bnp[2]["bias"], False)
    close16(out, c16.r16(o_ref + x, dtype), dtype, "out (BN3 + shortcut)")

    # ---- backward, stage by stage
    d3, dg3, db3 = c16.bn_bwd(dout, y3, m3, i3, bnp[2]["weight"], None, False, train)
    close16(dy3, c16.r16(d3, dtype), dtype, "dy3 (BN3 backward)")
    close32(grads[pfx + "point_linear.bn.bn.weight"], dg3, "dgamma3")
    close32(grads[pfx + "point_linear.bn.bn.bias"], db3, "dbeta3")
    a2_16 = c16.r16(a2, dtype)
    da2, dw2 = ora.pwconv_bwd(dy3, a2_16, c16.r16(w2, dtype))
    close32(grads[pfx + "point_linear.conv.conv.weight"], dw2, "dw2")
    assert np.all(grads[pfx + "point_linear.conv.conv.weight"][:, mid:] == 0)
    margin = 4 * RT[dtype] * 0.02
    safe2, safe1 = c16.edge_safe(pre2, margin), c16.edge_safe(pre1, margin)
    d2, dg2, db2 = c16.bn_bwd(c16.r16(da2, dtype), y2, m2, i2, bnp[1]["weight"], pre2, True, train)
    # where the depthwise weight gradient runs on the matrix cores with both operands formed on read (dw_wgrad_mfma_kernel
    # <..., true, true>) dy2 is stored nowhere: tA keeps da2 for that kernel, dy1 goes to tC
    wg_bx = any("dw_wgrad_mfma_kernel" in k and k.rstrip().endswith("true, true>") for k in bwd_table)
    if wg_bx:
        close16(tA, c16.r16(da2, dtype), dtype, "da2 (project dgrad, kept for the depthwise weight gradient)")
        d2g, _, _ = c16.bn_bwd(tA, y2, m2, i2, bnp[1]["weight"], pre2, True, train)
        dy2_gpu = c16.r16(d2g, dtype)          # dy2 as both depthwise kernels form it from the stored (da2, y2)
    else:
        dy2_gpu = tC if folded_bwd else tA
        close16(dy2_gpu, c16.r16(d2, dtype), dtype, "dy2 (project dgrad + BN2 backward)", safe2, 3e-4)
    sc = float(np.abs(da2).sum(axis=(0, 2, 3)).max())
    assert np.abs(grads[pfx + "depth_conv.bn.bn.weight"][:mid] - dg2).max() <= 2e-4 * sc * 3
    assert np.abs(grads[pfx + "depth_conv.bn.bn.bias"][:mid] - db2).max() <= 2e-4 * sc
    a1_in = c16.r16(a1, dtype) if dw_mma else a1.astype(np.float32)
    # the depthwise input gradient convolves the dy2 it forms itself: the stored 16-bit value on the matrix cores, the
    # fp32 value in the vector kernel (folded path); after an apply pass it reads the stored tensor
    dy2_in = (d2g if wg_bx else d2).astype(np.float32) if (folded_bwd and not dw_mma) else dy2_gpu
    da1, _ = ora.dwconv_bwd(dy2_in, a1_in, c16.r16(f_gpu, dtype) if dw_mma else f_gpu)
    # the weight-gradient kernel on the matrix cores sees the activated operand rounded to 16 bits; the vector kernel reads
    # it in fp32
    wg_mma = ran(bwd_table, "dw_wgrad_mfma_kernel") > 0
    _, df = ora.dwconv_bwd(dy2_gpu, c16.r16(a1, dtype) if wg_mma else a1.astype(np.float32), f_gpu)
    dwdw_ref, dm_ref = ora.ktransform_bwd(df, wdw, mid, K, ks_set, mats)
    close32(grads[pfx + "depth_conv.conv.conv.weight"], dwdw_ref, "d(depthwise weight)")
    for nm in ("7to5", "5to3"):
        gm = grads[pfx + "depth_conv.conv.%s_matrix" % nm]
        assert (gm is None) == (nm not in dm_ref), "None-ness of d(%s_matrix)" % nm
        if gm is not None:
            close32(gm, dm_ref[nm], "d(%s_matrix)" % nm)
    if folded_bwd:
        # tB = da1.  With the vector kernel the oracle's dy2 comes from its own da2: a flipped ReLU6 bit of BN2 (pre2
        # within round-off of a window edge) moves da1 in a k x k neighbourhood -- compare away from those
        near = np.zeros(tB.shape, bool)
        if not dw_mma:
            near = ora.dwconv_fwd((~safe2).astype(np.float32), np.ones((mid, 1, K, K), np.float32)) > 0
        close16(tB, c16.r16(da1, dtype), dtype, "da1 (depthwise dgrad of the folded BN2 backward)", ~near, 3e-4)
        d1, dg1, db1 = c16.bn_bwd(tB, y1, m1, i1, bnp[0]["weight"], pre1, True, train)
        if ran(bwd_table, "pw_wgrad_direct_kernel<%s, 3>" % T):
            # dy1 is stored nowhere: the expand input gradient (pw_fanin_pipe_kernel<..., BX>) and the expand weight
            # gradient (pw_wgrad_direct_kernel<T, 3>) both form it from the stored (da1, y1) as they read them
            dy1_gpu = c16.r16(d1, dtype)
        else:
            dy1_gpu = tC if wg_bx else tA
            close16(dy1_gpu, c16.r16(d1, dtype), dtype, "dy1 (BN1 backward, stored by the expand dgrad)", safe1, 3e-4)
    else:
        d1, dg1, db1 = c16.bn_bwd(c16.r16(da1, dtype), y1, m1, i1, bnp[0]["weight"], pre1, True, train)
        close16(tB, c16.r16(d1, dtype), dtype, "dy1 (depthwise dgrad + BN1 backward)", safe1, 3e-4)
        dy1_gpu = tB
    sc = float(np.abs(da1).sum(axis=(0, 2, 3)).max())
    assert np.abs(grads[pfx + "inverted_bottleneck.bn.bn.weight"][:mid] - dg1).max() <= 2e-4 * sc * 3
    assert np.abs(grads[pfx + "inverted_bottleneck.bn.bn.bias"][:mid] - db1).max() <= 2e-4 * sc
    dxe, dw1 = ora.pwconv_bwd(dy1_gpu, x, c16.r16(w1, dtype))
    close32(grads[pfx + "inverted_bottleneck.conv.conv.weight"], dw1, "dw1")
    assert np.all(grads[pfx + "inverted_bottleneck.conv.conv.weight"][mid:] == 0)
    close16(H(xg.grad), c16.r16(dxe.astype(np.float64) + dout, dtype), dtype, "dx (expand dgrad + shortcut)")
    for i, nm in enumerate(("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.")):
        assert np.all(grads[pfx + nm + "weight"][mid:] == 0) and np.all(grads[pfx + nm + "bias"][mid:] == 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("train", [True, False])
def test_composite_block_16bit_end_to_end_vs_fp32_oracle(dtype, train):
    """the whole 16-bit block call against the fp32 network oracle's block (oracle/s4_port.py:_mb_block, double) on the
    same 16-bit-rounded x / dout: relative L2 error of y, dx and every parameter gradient.  These are the 16-bit
    realisation errors of a 3-conv / 3-BN chain with two ReLU6 masks (a 16-bit rounding of the pre-activation flips
    masks near the window edges), not parity bars -- those are the stage checks above."""
    from oracle import s4_port
    ops = amd("ops")
    N, Hh, Ww, e, K = 4, 64, 64, 6, 7
    block, layer = _make_block(77)
    block.to(DEV).train(train)
    layer.active_kernel_size, layer.active_expand_ratio = K, e
    sd = {"blocks.0." + k: v.detach().cpu().double().clone() for k, v in block.state_dict().items()}
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
    g = torch.Generator().manual_seed(11)
    x16 = torch.randn((N, 64, Hh, Ww), generator=g).to(dtype)
    do16 = (0.05 * torch.randn((N, 64, Hh, Ww), generator=g)).to(dtype)
    xr = x16.double().requires_grad_(True)
    yr = s4_port._mb_block(sd, "blocks.0.mobile_inverted_conv.", xr, K, e, [3, 5, 7], True, train, 0.1, 1e-5)
    yr.backward(do16.double())
    xg = x16.to(DEV).requires_grad_(True)
    y = block(xg)
    y.backward(do16.to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / (b.norm() + 1e-30))

    bf = dtype == torch.bfloat16
    got = {"y": rel(y, yr.detach()), "dx": rel(xg.grad.detach(), xr.grad)}
    for n, p in block.named_parameters():
        r = sd["blocks.0." + n].grad
        assert (p.grad is None) == (r is None), n
        if r is not None:
            got[n] = rel(p.grad, r)
    print("16-bit realisation error (relative L2) %s %s: %s" % (dtype, "train" if train else "eval",
                                                                 {k: round(v, 5) for k, v in got.items()}))
    # measured on MI355X (gpurun_out/r2_t2.log): bf16 train y 0.4 %, dx 4.5 %, gradients <= 5 %; f16 8x smaller
    # measured (gpurun_out/r2_t3b.log): bf16 train y 0.46 %, dx 4.5 %, parameter gradients <= 8.1 %; f16 train y 0.06 %,
    # dx 1.6 %, <= 2.8 %; eval-mode BN about 0.6x of those
    bound = {"y": 6e-3 if bf else 8e-4, "dx": 7e-2 if bf else 2.5e-2}
    for k, v in got.items():
        assert v <= bound.get(k, 1.2e-1 if bf else 4.5e-2), (k, v, got)
