"""GPU parity of the fused BatchNorm(+ReLU6)(+residual) kernels (C ABI ofasr_bn_*) against the CPU oracle's
sliced BatchNorm (oracle/ofasr_oracle.c ora_bn_*, pinned to the reference by tests/golden/bn.npz)."""
import numpy as np
import pytest
import torch

from conftest import amd, assert_close
from detfill import det_uniform, fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bn(cmax, momentum=0.1, prefix="bnfix"):
    bn = torch.nn.BatchNorm2d(cmax, momentum=momentum, eps=1e-5)
    sd = fill_state_dict({"weight": (cmax,), "bias": (cmax,), "running_mean": (cmax,), "running_var": (cmax,)}, prefix)
    bn.weight.data.copy_(torch.from_numpy(sd["weight"]))
    bn.bias.data.copy_(torch.from_numpy(sd["bias"]))
    bn.running_mean.copy_(torch.from_numpy(sd["running_mean"]))
    bn.running_var.copy_(torch.from_numpy(sd["running_var"]))
    return bn, sd


def _tol(dtype):
    return {torch.float32: (5e-5, 5e-6), torch.bfloat16: (1.5e-2, 1.5e-2), torch.float16: (3e-3, 3e-3)}[dtype]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("shape", [(3, 16, 5, 6), (2, 24, 8, 8), (4, 7, 3, 5)])
def test_bn_act_vs_oracle(ora, dtype, training, act, res, shape):
    ops = amd("ops")
    N, C, H, W = shape
    cmax = 24
    bn, sd = _bn(cmax)
    bn.to(DEV).train(training)
    x = torch.from_numpy(det_uniform(shape, "bna/x%s" % (shape,), -2.0, 2.0)).to(dtype)
    r = torch.from_numpy(det_uniform(shape, "bna/r%s" % (shape,), -1.0, 1.0)).to(dtype) if res else None
    dy = torch.from_numpy(det_uniform(shape, "bna/dy%s" % (shape,))).to(dtype)
    xg = x.to(DEV).requires_grad_(True)
    rg = r.to(DEV).requires_grad_(True) if res else None
    y = ops.bn_act(xg, bn, act, rg)
    # oracle on the same (rounded) inputs
    rm, rv = sd["running_mean"].copy(), sd["running_var"].copy()
    xf = x.float().numpy()
    yb, mean, invstd = ora.bn_fwd(xf, sd["weight"], sd["bias"], rm, rv, training)
    pre = yb + (r.float().numpy() if res else 0.0)
    y_ref = np.clip(pre, 0.0, 6.0) if act else pre
    rt, at = _tol(dtype)
    assert_close(y.detach().float().cpu().numpy(), y_ref, rt, at, "y")
    if training:
        assert_close(bn.running_mean.cpu().numpy(), rm, 1e-5, 1e-6, "running_mean")
        assert_close(bn.running_var.cpu().numpy(), rv, 1e-5, 1e-6, "running_var")
        assert int(bn.num_batches_tracked) == 1
        assert np.array_equal(bn.running_mean.cpu().numpy()[C:], sd["running_mean"][C:])
    y.backward(dy.to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    dyf = dy.float().numpy()
    dz = dyf * ((pre > 0) & (pre < 6)) if act else dyf
    if training:
        dx_ref, dg_ref, db_ref = ora.bn_bwd_train(dz, xf, sd["weight"])
    else:
        g = sd["weight"][:C].reshape(1, C, 1, 1)
        istd = (1.0 / np.sqrt(sd["running_var"][:C].astype(np.float64) + 1e-5)).reshape(1, C, 1, 1)
        mu = sd["running_mean"][:C].astype(np.float64).reshape(1, C, 1, 1)
        dx_ref = (dz * g * istd).astype(np.float32)
        dg_ref = (dz * (xf - mu) * istd).sum(axis=(0, 2, 3)).astype(np.float32)
        db_ref = dz.sum(axis=(0, 2, 3)).astype(np.float32)
    # near the ReLU6 window edges a 16-bit rounding of `pre` may flip the mask: compare away from the edges
    safe = np.ones_like(pre, bool)
    if act and dtype != torch.float32:
        safe = (np.abs(pre) > 0.05) & (np.abs(pre - 6) > 0.05)
    if dtype == torch.float32 or not training:
        got = xg.grad.float().cpu().numpy()
        assert_close(np.where(safe, got, 0), np.where(safe, dx_ref, 0), 5 * rt, 5 * at, "dx")
    assert_close(bn.weight.grad.cpu().numpy()[:C], dg_ref, 5 * rt, 20 * at, "dgamma")
    assert_close(bn.bias.grad.cpu().numpy()[:C], db_ref, 5 * rt, 20 * at, "dbeta")
    assert np.all(bn.weight.grad.cpu().numpy()[C:] == 0) and np.all(bn.bias.grad.cpu().numpy()[C:] == 0)
    if res:
        assert_close(np.where(safe, rg.grad.float().cpu().numpy(), 0), np.where(safe, dz, 0), 1e-6, 1e-6, "dres")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(16, 48, 64, 64), (5, 64, 64, 64), (16, 96, 32, 32), (3, 48, 96, 96)])
def test_bn_bwd_large_shapes(dtype, act, training, shape):
    """BASELINE-sized channel slices (16 x 64 x 64 per channel and ragged variants): dx, dgamma, dbeta of the 16-bit
    backward against an fp64 restatement of the same formulas on the same 16-bit inputs."""
    ops = amd("ops")
    N, C, H, W = shape
    bn, sd = _bn(C, prefix="bnres")
    bn.to(DEV).train(training)
    g = torch.Generator().manual_seed(1234 + N + C)
    x = (torch.randn(shape, generator=g) * 1.5).to(dtype)
    dy = torch.randn(shape, generator=g).to(dtype)
    xg = x.to(DEV).requires_grad_(True)
    rm0, rv0 = bn.running_mean.double().cpu().clone(), bn.running_var.double().cpu().clone()
    y = ops.bn_act(xg, bn, act)
    y.backward(dy.to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    xd, dyd = x.double(), dy.double()
    if training:
        mu = xd.mean(dim=(0, 2, 3), keepdim=True)
        var = xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mu, var = rm0.view(1, C, 1, 1), rv0.view(1, C, 1, 1)
    istd = 1.0 / torch.sqrt(var + 1e-5)
    gam = bn.weight.detach().double().cpu().view(1, C, 1, 1)
    bet = bn.bias.detach().double().cpu().view(1, C, 1, 1)
    xhat = (xd - mu) * istd
    pre = xhat * gam + bet
    dz = dyd * ((pre > 0) & (pre < 6)) if act else dyd
    db = dz.sum(dim=(0, 2, 3))
    dg = (dz * xhat).sum(dim=(0, 2, 3))
    M = N * H * W
    dx = gam * istd * (dz - (db / M).view(1, C, 1, 1) - xhat * (dg / M).view(1, C, 1, 1)) if training else dz * gam * istd
    safe = torch.ones_like(pre, dtype=torch.bool)
    if act:
        safe = (pre.abs() > 0.05) & ((pre - 6).abs() > 0.05)
    rt, at = _tol(dtype)
    got = xg.grad.double().cpu()
    assert_close(torch.where(safe, got, 0.0).numpy(), torch.where(safe, dx, 0.0).numpy(), rt, at, "dx")
    # the sums run over 1e4..1e5 terms of 16-bit data; a flipped mask bit at a window edge moves them by one term
    scale = float(dz.abs().sum(dim=(0, 2, 3)).max())
    assert float((bn.bias.grad.double().cpu() - db).abs().max()) <= 2e-4 * scale + 0.2 * act
    assert float((bn.weight.grad.double().cpu() - dg).abs().max()) <= 2e-4 * scale * 3 + 0.6 * act


def test_bn_golden_reference(golden):
    """the same goldens that pin the oracle (reference DynamicBatchNorm2d fwd/bwd, train and eval)."""
    dop = amd("elastic_nn.modules.dynamic_op")
    g = golden("bn.npz")
    for C in (16, 24):
        for training in (True, False):
            m = dop.DynamicBatchNorm2d(24)
            m.bn.momentum, m.bn.eps = 0.1, 1e-5
            sd = fill_state_dict({"bn.weight": (24,), "bn.bias": (24,), "bn.running_mean": (24,),
                                  "bn.running_var": (24,)}, "bnfix")
            m.bn.weight.data.copy_(torch.from_numpy(sd["bn.weight"]))
            m.bn.bias.data.copy_(torch.from_numpy(sd["bn.bias"]))
            m.bn.running_mean.copy_(torch.from_numpy(sd["bn.running_mean"]))
            m.bn.running_var.copy_(torch.from_numpy(sd["bn.running_var"]))
            m.to(DEV).train(training)
            x = torch.from_numpy(det_uniform((3, C, 5, 6), "bn/x%d" % C, -2.0, 2.0)).to(DEV).requires_grad_(True)
            y = m(x)
            tag = "c%d_%s" % (C, "train" if training else "eval")
            assert_close(y.detach().cpu().numpy(), g["y_" + tag], 5e-5, 5e-6, "y")
            y.backward(torch.from_numpy(det_uniform(tuple(y.shape), "bn/dy%d" % C)).to(DEV))
            amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
            assert_close(x.grad.cpu().numpy(), g["dx_" + tag], 1e-4, 1e-5, "dx")
            assert_close(m.bn.weight.grad.cpu().numpy(), g["dgamma_" + tag], 5e-5, 5e-5, "dgamma")
            assert_close(m.bn.bias.grad.cpu().numpy(), g["dbeta_" + tag], 5e-5, 5e-5, "dbeta")
            assert_close(m.bn.running_mean.cpu().numpy(), g["rm_" + tag], 1e-5, 1e-6, "rm")
            assert_close(m.bn.running_var.cpu().numpy(), g["rv_" + tag], 1e-5, 1e-6, "rv")
            assert int(m.bn.num_batches_tracked) == int(g["nbt_" + tag])


def test_fused_and_modular_paths_agree():
    ops = amd("ops")
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = amd("elastic_nn.networks").OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6],
                                                    depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    torch.manual_seed(0)
    net.init_model("he_fout")
    net.to(DEV).train()
    net.set_active_subnet(ks=5, e=4, d=3, pixel_d=2)
    x = torch.rand(2, 3, 16, 12, device=DEV)
    target = torch.rand(2, 3, 64, 48, device=DEV)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    outs = []
    for fused in (True, False):
        net.load_state_dict(sd)
        net.zero_grad()
        ops.FUSED_BN = fused
        try:
            y = net(x)
            # NB not mean(y^2): y is a train-mode BN output, whose mean square is constant => zero gradient, pure noise
            (y - target).square().mean().backward()
            amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
        finally:
            ops.FUSED_BN = True
        outs.append((y.detach().clone(), net.blocks[0].mobile_inverted_conv.depth_conv.conv.conv.weight.grad.clone(),
                     net.blocks[0].mobile_inverted_conv.depth_conv.bn.bn.running_var.clone()))
    for a, b in zip(outs[0], outs[1]):
        rel = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        assert rel < 2e-2, rel


def test_composite_block_matches_per_op_path():
    """ofasr_mbconv_fwd/_bwd (one FFI call per block and direction) vs the per-op autograd Functions.  fp32: same kernels
    in the same order (statistics from the conv epilogues in the composite call), outputs and every gradient agree to 1e-4.  16-bit: the composite call applies BN1/BN2 + ReLU6
    inside the consumer kernels from the pre-BN tensor (the activated tensor is never rounded to 16 bits), so it is a
    different -- not a worse -- 16-bit realisation: both are measured against the fp32 result in the L2 norm
    (tools/chk_fused.py prints the numbers: 4.7 % vs 4.6 % on dx in train mode, 0.4 % on y)."""
    ops = amd("ops")
    dop = amd("elastic_nn.modules.dynamic_op")
    dl = amd("elastic_nn.modules.dynamic_layers")
    blk = amd("imagenet_codebase.networks")
    layers = amd("layers")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1

    def run(block, sd, x0, dy, dtype, composite):
        block.load_state_dict(sd)
        block.zero_grad()
        ops.FUSED_BLOCK = composite
        try:
            x = x0.to(dtype).clone().requires_grad_(True)
            y = block(x)
            y.backward(dy.to(dtype))
            amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
        finally:
            ops.FUSED_BLOCK = True
        return (y.detach().float(), x.grad.float(),
                {n: (None if p.grad is None else p.grad.clone()) for n, p in block.named_parameters()},
                {n: b.clone() for n, b in block.named_buffers()})

    for (k, e, train) in [(7, 6, True), (5, 4, True), (3, 3, False)]:
        torch.manual_seed(3)
        layer = dl.DynamicMBConvLayer([64], [64], [3, 5, 7], [3, 4, 6])
        block = blk.MobileInvertedResidualBlock(layer, layers.IdentityLayer([64], [64])).to(DEV).train(train)
        for m in block.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
        layer.depth_conv.conv.__getattr__("7to5_matrix").data.add_(0.1 * torch.randn(25, 25, device=DEV))
        layer.active_kernel_size, layer.active_expand_ratio = k, e
        x0 = torch.randn(2, 64, 16, 24, device=DEV).bfloat16().float()
        dy = torch.randn(2, 64, 16, 24, device=DEV).bfloat16().float()
        sd = {kk: v.clone() for kk, v in block.state_dict().items()}
        (ya, xa, ga, ba) = run(block, sd, x0, dy, torch.float32, True)
        (yb, xb, gb, bb) = run(block, sd, x0, dy, torch.float32, False)
        # same kernels in the same order, except the batch statistics: the composite call takes them from the conv
        # kernels' epilogues (fp32 (sum, sum of squares) per tile, fp64 fold), the per-op path from the fp64 statistics
        # pass -- 2e-5 on a BN weight gradient (a difference of large sums)
        tol = 1e-4
        assert float((ya - yb).abs().max()) <= tol * max(1.0, float(yb.abs().max()))
        assert float((xa - xb).abs().max()) <= tol * max(1.0, float(xb.abs().max()))
        for n in ga:
            assert (ga[n] is None) == (gb[n] is None), n
            if ga[n] is not None:
                # (BN parameter gradients are sums of O(N*H*W) = 768 signed O(1) terms that nearly cancel: their round-off
                # is absolute, ~768 * 2^-24 per re-ordering, whatever the size of the sum)
                assert float((ga[n] - gb[n]).abs().max()) <= tol * max(1.0 if ".bn." in n else 1e-3, float(gb[n].abs().max())), n
        for n in ba:
            assert torch.allclose(ba[n].float(), bb[n].float(), rtol=5e-5, atol=2e-6), n
        # 16-bit: composite (fused BN apply) and per-op, each against the fp32 per-op result
        (yc, xc, gc, bc) = run(block, sd, x0, dy, torch.bfloat16, True)
        (yd, xd, gd, bd) = run(block, sd, x0, dy, torch.bfloat16, False)
        rel = lambda a, r: float((a - r).norm()) / max(float(r.norm()), 1e-12)
        assert rel(yc, yb) <= 1.25 * rel(yd, yb) + 1e-3 and rel(yc, yb) <= 1e-2
        assert rel(xc, xb) <= 1.25 * rel(xd, xb) + 1e-3
        for n in gb:
            assert (gc[n] is None) == (gb[n] is None), n
            if gb[n] is not None:
                assert rel(gc[n], gb[n]) <= 1.25 * rel(gd[n], gb[n]) + 2e-3, n
        for n in bb:
            assert torch.allclose(bc[n].float(), bb[n].float(), rtol=2e-2, atol=2e-3), n
