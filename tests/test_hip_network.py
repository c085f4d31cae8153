"""GPU parity of the host mirror (DynamicMBConvLayer, OFAMobileNetS4) running on the HIP kernels,
against goldens produced by the reference itself and against the CPU oracle.  `-m gpu`.

Bar (BASELINE.json north_star): same weights + same inputs => |dPSNR| <= 1e-3 dB in fp32;
tensor-level tolerances below are fp32 reassociation budgets over a 20-60 layer deep net.
"""
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import GOLDEN, amd, assert_close
from detfill import det_uniform, fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def H(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLDEN, "s4_meta.json")))


@pytest.fixture(scope="module")
def mods():
    assert torch.cuda.is_available()
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    return dict(dop=dop, dl=amd("elastic_nn.modules.dynamic_layers"), nets=amd("elastic_nn.networks"),
                blk=amd("imagenet_codebase.networks"), layers=amd("layers"), utils=amd("utils"))


def _load(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = fill_state_dict(shapes, prefix)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


@pytest.mark.parametrize("bn_train", [True, False])
@pytest.mark.parametrize("ke", [(7, 6), (5, 4), (3, 3), (3, 6), (7, 3)])
def test_mb_block_golden(mods, golden, bn_train, ke):
    k, e = ke
    g = golden("mbblock.npz")
    C = 16
    layer = mods["dl"].DynamicMBConvLayer([C], [C], [3, 5, 7], [3, 4, 6], stride=1, act_func="relu6")
    block = mods["blk"].MobileInvertedResidualBlock(layer, mods["layers"].IdentityLayer([C], [C]))
    for m in block.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum, m.eps = 0.1, 1e-5
    _load(block, "mbblock")
    block.to(DEV).train(bn_train)
    layer.active_kernel_size, layer.active_expand_ratio = k, e
    x = G(g["x"]).requires_grad_(True)
    y = block(x)
    tag = "k%d_e%d_%s" % (k, e, "train" if bn_train else "eval")
    assert_close(H(y), g["y_" + tag], 1e-4, 1e-5, "y")
    y.backward(G(det_uniform(tuple(y.shape), "mb/dy")))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert_close(H(x.grad), g["dx_" + tag], 2e-4, 2e-5, "dx")
    for name, p in block.named_parameters():
        assert bool(g["isnone_%s_%s" % (name, tag)]) == (p.grad is None), name
        if p.grad is not None:
            ref = g["grad_%s_%s" % (name, tag)]
            assert_close(H(p.grad), ref, 5e-4, 2e-5 * max(1.0, float(np.abs(ref).max())), name)
    if bn_train:
        for name, b in block.named_buffers():
            assert_close(H(b), g["buf_%s_%s" % (name, tag)], 1e-5, 1e-6, name)


def _make_s4(mods, meta):
    net = mods["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                      pixelshuffle_depth_list=[1, 2])
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["state_dict_shapes"]
    assert [n for n, _ in net.named_parameters()] == meta["param_names"]
    _load(net, "s4")
    return net.to(DEV)


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("bn_train", [True, False])
def test_s4_golden(mods, golden, meta, si, bn_train):
    g = golden("s4_net.npz")
    net = _make_s4(mods, meta)
    net.train(bn_train)
    net.set_active_subnet(**meta["settings"][si])
    if bn_train:
        assert net.runtime_depth == meta["runtime_depth"][si]
    y = net(G(g["lr"]))
    tag = "s%d_%s" % (si, "train" if bn_train else "eval")
    assert list(y.shape) == meta["out_shapes"][si]
    assert_close(H(y), g["y_" + tag], 5e-4, 5e-5, "y")
    hr = G(det_uniform(tuple(y.shape), "s4/hr%d" % si, 0.0, 1.0))
    loss = F.mse_loss(y, hr)
    assert abs(float(loss) - float(g["loss_" + tag])) <= 2e-5 * abs(float(g["loss_" + tag]))
    loss.backward()
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    names = meta["param_names"]
    params = dict(net.named_parameters())
    isnone = np.array([params[n].grad is None for n in names])
    assert np.array_equal(isnone, g["g_isnone_" + tag])
    l2 = np.array([0.0 if params[n].grad is None else float(params[n].grad.double().pow(2).sum().sqrt())
                   for n in names])
    assert_close(l2, g["g_l2_" + tag], 5e-3, 1e-7, "grad l2 norms")
    for k in g.files:
        if k.startswith("grad_") and k.endswith("_" + tag):
            name = k[len("grad_"):-len("_" + tag)]
            ref = g[k]
            # atol: fp32 round-off accumulated over a ~60-layer backward chain (train-mode BN differences cancel)
            assert_close(H(params[name].grad), ref, 5e-3, 2e-5 * max(1.0, float(np.abs(ref).max())), name)
    if bn_train:
        bufs = dict(net.named_buffers())
        for k in g.files:
            if k.startswith("buf_") and k.endswith("_" + tag):
                name = k[len("buf_"):-len("_" + tag)]
                assert_close(H(bufs[name]), g[k], 2e-5, 2e-6, name)


def test_s4_psnr_parity_fp32(mods, golden, meta):
    """north_star: |dPSNR| <= 1e-3 dB between the reference (CPU/PyTorch) and the HIP path on identical
    weights and inputs, through the reference's own metric definition."""
    g = golden("s4_net.npz")
    net = _make_s4(mods, meta).eval()
    net.set_active_subnet(ks=7, e=6, d=4, pixel_d=2)
    with torch.no_grad():
        y1 = net(G(g["lr"][:1]))
    assert_close(H(y1), g["psnr_y1"], 5e-4, 5e-5, "y1")
    val = mods["utils"].psnr_y(y1, torch.from_numpy(g["psnr_tgt"]))
    assert abs(val - float(g["psnr_value"])) <= 1e-3, (val, float(g["psnr_value"]))


def test_s4_sampling_and_constraints(mods, meta):
    net = _make_s4(mods, meta)
    for t in meta["sample_traces"]:
        random.seed(t["seed"])
        s = net.sample_active_subnet()
        assert s == t["sampled"] and net.runtime_depth == t["runtime_depth"]
        assert [b.mobile_inverted_conv.active_kernel_size for b in net.blocks[:-2]] == t["ks"]
    net.set_constraint([4, 3], constraint_type="depth")
    net.set_constraint([7, 5], constraint_type="kernel_size")
    random.seed(meta["constrained_trace"]["seed"])
    s = net.sample_active_subnet()
    assert s == meta["constrained_trace"]["sampled"]
    assert net.runtime_depth == meta["constrained_trace"]["runtime_depth"]
    net.clear_constraint()


def test_s4_vs_oracle_random_subnet_and_bf16(mods, meta):
    """a sampled sub-network at a size no golden covers, checked against the network-level CPU oracle
    (fp32), then the same step under bf16 autocast (activation dtype of the bench) with a loose bound."""
    from oracle import s4_port
    net = _make_s4(mods, meta).train()
    random.seed(1234)
    sampled = net.sample_active_subnet()
    arch = s4_port.Arch()
    random.seed(1234)
    assert arch.sample_active_subnet() == sampled
    lr = det_uniform((2, 3, 24, 16), "s4o/lr", 0.0, 1.0)
    sd = {k: torch.from_numpy(v.copy()) for k, v in
          fill_state_dict({k: tuple(v) for k, v in meta["state_dict_shapes"].items()}, "s4").items()}
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
    y_ref = s4_port.s4_forward(sd, torch.from_numpy(lr), arch, training=True)
    hr = torch.from_numpy(det_uniform(tuple(y_ref.shape), "s4o/hr", 0.0, 1.0))
    F.mse_loss(y_ref, hr).backward()
    y = net(G(lr))
    assert_close(H(y), y_ref.detach().numpy(), 5e-4, 5e-5, "y")
    F.mse_loss(y, hr.to(DEV)).backward()
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    # NB this det-filled net is chaotic in backward: perturbing the INPUT by 1e-7 (relative) moves some gradients
    # of the first stage by ~5 % of their max through a ReLU6 mask flip (measured on the ATen path itself), so
    # implementations with different but equally valid rounding can only be compared robustly here; the
    # element-wise gradient parity is pinned by the golden tests above, whose inputs sit away from such flips.
    for name, p in net.named_parameters():
        r = sd[name].grad
        assert (p.grad is None) == (r is None), name
        if r is not None:
            a, b = p.grad.detach().cpu().double().flatten(), r.double().flatten()
            rel = float((a - b).norm() / (b.norm() + 1e-30))
            assert rel < 0.15, (name, rel)
    tot_a = torch.cat([p.grad.detach().cpu().double().flatten() for _, p in net.named_parameters() if p.grad is not None])
    tot_b = torch.cat([sd[n].grad.double().flatten() for n, p in net.named_parameters() if p.grad is not None])
    assert float((tot_a - tot_b).norm() / tot_b.norm()) < 2e-2


def test_s4_bf16_whole_net_gradient_error_vs_oracle(mods):
    """bf16 activations (fp32 master weights, fp32 accumulation) on the timed path's shapes: a he_fout-initialised S4
    supernet (the state a training run starts from; the det-filled net above is chaotic in backward), N=4, LR 64x64,
    sampled sub-network, train-mode BN -- output, loss and the WHOLE-NET relative gradient error
    ||g_hip - g_oracle|| / ||g_oracle|| over all 2.16 M parameters against the fp32 network oracle, plus the same per
    tensor for the tensors that carry 99 % of the gradient energy."""
    from oracle import s4_port
    torch.manual_seed(21)
    net = mods["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                      pixelshuffle_depth_list=[1, 2])
    net.init_model("he_fout")
    net.to(DEV).train()
    random.seed(77)
    sampled = net.sample_active_subnet()
    arch = s4_port.Arch()
    random.seed(77)
    assert arch.sample_active_subnet() == sampled
    g = torch.Generator().manual_seed(5)
    scale = net.active_upscale()
    hr = torch.rand((4, 3, 64 * scale, 64 * scale), generator=g)
    lr = F.interpolate(hr, scale_factor=1.0 / scale, mode="bicubic", antialias=True).clamp_(0, 1)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
    y_ref = s4_port.s4_forward(sd, lr, arch, training=True)
    loss_ref = F.mse_loss(y_ref, hr)
    loss_ref.backward()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = net(lr.to(DEV))
    assert yb.dtype == torch.bfloat16
    loss = F.mse_loss(yb.float(), hr.to(DEV))
    loss.backward()
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    rel_y = float((yb.float().cpu() - y_ref.detach()).norm() / y_ref.detach().norm())
    pairs = [(n, p.grad.detach().double().cpu().flatten(), sd[n].grad.double().flatten())
             for n, p in net.named_parameters() if p.grad is not None]
    for n, p in net.named_parameters():
        assert (p.grad is None) == (sd[n].grad is None), n
    ga, gb = torch.cat([a for _, a, _ in pairs]), torch.cat([b for _, _, b in pairs])
    whole = float((ga - gb).norm() / gb.norm())
    energy = sorted(((float(b.norm() ** 2), n, float((a - b).norm() / (b.norm() + 1e-30))) for n, a, b in pairs), reverse=True)
    tot, acc, worst = sum(e for e, _, _ in energy), 0.0, 0.0
    for e, n, r in energy:
        worst = max(worst, r)
        acc += e
        if acc >= 0.99 * tot:
            break
    print("bf16 S4: loss %.6f vs %.6f, output rel %.4g, whole-net gradient rel %.4g, worst dominant tensor %.4g"
          % (float(loss), float(loss_ref), rel_y, whole, worst))
    assert abs(float(loss) - float(loss_ref)) <= 2e-2 * float(loss_ref)
    assert rel_y <= 6e-2                       # 3.4 % measured on MI355X (bf16 storage of ~60 layers)
    assert whole <= 0.2 and worst <= 0.4


def test_get_active_subnet_matches_supernet(mods):
    C = 64
    layer = mods["dl"].DynamicMBConvLayer([C], [C], [3, 5, 7], [3, 4, 6]).to(DEV)
    _load(layer, "gas")
    layer.to(DEV).eval()
    layer.active_kernel_size, layer.active_expand_ratio = 5, 4
    sub = layer.get_active_subnet(C).eval()
    x = G(det_uniform((2, C, 20, 24), "gas/x"))
    with torch.no_grad():
        a, b = layer(x), sub(x)
    assert_close(H(b), H(a), 1e-5, 1e-6, "static sub-layer vs elastic layer")
    assert tuple(sub.depth_conv.conv.weight.shape) == (256, 1, 5, 5)


def test_bn_recalibration_golden(mods, golden):
    """set_running_statistics (reference ofa/elastic_nn/utils.py:16-64) on the sub-network (ks=5, e=4, d=3, pixel_d=2):
    running statistics of every BatchNorm after two calibration batches of different size, against the reference's
    own result (tests/golden/calibration.npz, make_golden.py gen_calibration)."""
    g = golden("calibration.npz")
    eutils = amd("elastic_nn.utils")
    net = mods["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                      pixelshuffle_depth_list=[1, 2])
    _load(net, "s4")
    net = net.to(DEV).eval()
    net.set_active_subnet(ks=5, e=4, d=3, pixel_d=2)
    before = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k}
    loader = [{"image": G(g["b0"])}, {"image": G(g["b1"])}]
    eutils.set_running_statistics(net, loader)
    changed = 0
    for k, v in net.state_dict().items():
        if "running_mean" in k or "running_var" in k:
            ref = g[k]
            assert_close(H(v), ref, 2e-4, 2e-5 * max(1.0, float(np.abs(ref).max())), k)
            changed += int(not torch.equal(v, before[k]))
    assert changed > 40   # the active BNs were re-estimated, inactive ones (skipped depth) keep their values


def test_recalibration_invalidates_inference_operands(mods, golden):
    """set_running_statistics writes the new statistics through `.data` (no version bump), so the
    BN-folded operand cache and GraphedEval's captured graphs must be dropped by it.  bf16 eval forward ->
    re-calibration -> eval forward must equal the same sequence with the operand cache switched off, and GraphedEval
    must capture again."""
    g = golden("calibration.npz")
    eutils, ops = amd("elastic_nn.utils"), amd("ops")
    graphed = amd("graphed")

    def run(cache):
        was = ops.INFER_CACHE
        ops.INFER_CACHE = cache
        ops.clear_infer_cache()
        try:
            net = mods["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                              pixelshuffle_depth_list=[1, 2])
            _load(net, "s4")
            net = net.to(DEV).eval()
            net.set_active_subnet(ks=5, e=4, d=3, pixel_d=2)
            x = G(g["b0"])[:2]
            ge = graphed.GraphedEval(net, autocast_dtype=torch.bfloat16)
            with torch.no_grad():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y0 = net(x).float().clone()
                z0 = ge(x).float().clone()
                c0 = ge.captures
                eutils.set_running_statistics(net, [{"image": G(g["b0"])}, {"image": G(g["b1"])}])
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y1 = net(x).float().clone()
                z1 = ge(x).float().clone()
            stats = torch.cat([v.flatten().float() for k, v in net.state_dict().items() if "running_" in k])
            return y0, y1, z0, z1, ge.captures - c0, stats
        finally:
            ops.INFER_CACHE = was
            ops.clear_infer_cache()

    y0, y1, z0, z1, recaptured, st = run(True)
    u0, u1, _, _, _, su = run(False)
    assert torch.equal(y0, u0) and torch.equal(z0, y0)
    # the calibration forward itself (per-op fp32 path + ATen reductions) repeats to within an ulp of the statistics
    assert float(((st - su).abs() / (1.0 + su.abs())).max()) <= 1e-5, \
        "re-calibrated statistics differ: max %g" % float((st - su).abs().max())
    scale = float(u1.abs().max())
    moved = float((y0 - y1).abs().max())
    stale = float((y1 - u1).abs().max())
    assert moved >= 0.05 * scale, "re-calibration must change the eval output (moved %g of %g)" % (moved, scale)
    # operands folded from the OLD statistics would reproduce y0, i.e. be off by `moved`; an ulp of the statistics is
    # at most one bf16 rounding step of the output
    assert stale <= 0.02 * scale and stale <= 0.1 * moved, \
        "cached inference operands survived set_running_statistics (max diff %g, re-calibration moved %g)" % (stale, moved)
    assert recaptured == 1 and float((z1 - y1).abs().max()) == 0.0, "GraphedEval replayed a graph of the old statistics"
