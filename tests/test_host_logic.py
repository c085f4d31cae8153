"""CPU tests of the host-side mirror: everything that does not launch a kernel -- module construction
and state-dict layout, sub-network control / sampling parity with the reference, optimizer grouping,
LR schedules, checkpoint key remapping, the PSNR metric.  (Forward passes need the GPU: test_hip_*.)"""
import json
import math
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN, amd, assert_close


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLDEN, "s4_meta.json")))


@pytest.fixture(scope="module")
def net():
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    n = amd("elastic_nn.networks").OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6],
                                                  depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    return n


def test_state_dict_layout_and_param_order(net, meta):
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["state_dict_shapes"]
    assert [n for n, _ in net.named_parameters()] == meta["param_names"]
    assert sum(p.numel() for p in net.parameters()) == meta["n_params"] == 2160422
    assert net.block_group_info == meta["block_group_info"]
    assert tuple(net.state_dict()["blocks.0.mobile_inverted_conv.depth_conv.conv.7to5_matrix"].shape) == (25, 25)
    assert tuple(net.state_dict()["blocks.16.conv.weight"].shape) == (256, 64, 5, 5)


def test_transform_params_only_when_mode_set_at_construction():
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    op = dop.DynamicSeparableConv2d(8, [3, 5, 7])
    assert [n for n, _ in op.named_parameters()] == ["conv.weight"]
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    op = dop.DynamicSeparableConv2d(8, [3, 5, 7])
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    assert sorted(n for n, _ in op.named_parameters()) == ["5to3_matrix", "7to5_matrix", "conv.weight"]
    assert torch.equal(getattr(op, "7to5_matrix"), torch.eye(25))
    assert op._chain(3) == (7, 5, 3) and op._chain(5) == (7, 5) and op._chain(7) == (7,)


def test_optimizer_groups_match_reference(net, meta):
    keys = ["bn", "bias"]
    assert len(list(net.get_parameters(keys, mode="exclude"))) == meta["n_decay"] == 86
    assert len(list(net.get_parameters(keys, mode="include"))) == meta["n_no_decay"] == 108


def test_sampling_traces_and_quirks(net, meta):
    for t in meta["sample_traces"]:
        random.seed(t["seed"])
        s = net.sample_active_subnet()
        assert s == t["sampled"]                      # incl. the mutated 'd' list (Q2)
        assert net.runtime_depth == t["runtime_depth"]
    for setting, rd in zip(meta["settings"], meta["runtime_depth"]):
        net.set_active_subnet(**setting)
        assert net.runtime_depth == rd
    # Q1: compat gates the shuffle stage with runtime_depth[0]
    net.set_active_subnet(ks=5, e=4, d=3, pixel_d=1)
    kinds = [k for k, _ in net.active_block_sequence()]
    assert kinds.count("shuffle") == 2 and kinds.count("mb") == 3 + 3 + 3 + 1
    type(net).COMPAT_REFERENCE_INDEXING = False
    try:
        net.set_active_subnet(ks=5, e=4, d=3, pixel_d=1)
        kinds = [k for k, _ in net.active_block_sequence()]
        assert kinds.count("shuffle") == 1 and kinds.count("mb") == 12
        assert net.runtime_depth == [3, 3, 3, 3, 1]
    finally:
        type(net).COMPAT_REFERENCE_INDEXING = True
    net.set_constraint([4, 3], constraint_type="depth")
    net.set_constraint([7, 5], constraint_type="kernel_size")
    random.seed(meta["constrained_trace"]["seed"])
    assert net.sample_active_subnet() == meta["constrained_trace"]["sampled"]
    net.clear_constraint()
    with pytest.raises(NotImplementedError):
        net.set_constraint([1], constraint_type="nope")


def test_make_divisible_and_mid_channels():
    u = amd("utils")
    assert [u.make_divisible(round(64 * e), 8) for e in (3, 4, 6)] == [192, 256, 384]
    assert u.make_divisible(10, 8) == 16 and u.make_divisible(3, 8) == 8 and u.make_divisible(91, 8) == 88
    assert u.sub_filter_start_end(7, 5) == (1, 6) and u.sub_filter_start_end(5, 3) == (1, 4)
    assert u.sub_filter_start_end(7, 3) == (2, 5)
    lst = [1, 2]
    assert u.int2list(lst) is lst and u.int2list(3, 2) == [3, 3] and u.int2list((1, 2)) == [1, 2]


def test_load_weights_from_net_key_remap(net):
    sd = net.state_dict()
    src = {}
    for k, v in sd.items():
        k2 = k.replace(".conv.conv.weight", ".conv.weight").replace(".bn.bn.", ".bn.")
        src["module." + k2] = torch.full_like(v, 0.5) if v.is_floating_point() else v
    net.load_weights_from_net(src)
    assert float(net.state_dict()["blocks.3.mobile_inverted_conv.point_linear.conv.conv.weight"].mean()) == 0.5
    with pytest.raises((ValueError, AssertionError)):
        net.load_weights_from_net({"nonsense.key": torch.zeros(1)})


def test_run_config_lr_schedule_and_optimizer(net):
    rm = amd("imagenet_codebase.run_manager")
    cfg = rm.SyntheticSRRunConfig(n_epochs=3, init_lr=1e-3)
    assert cfg.calc_learning_rate(0, 0, 10) == pytest.approx(1e-3)
    assert cfg.calc_learning_rate(1, 5, 10) == pytest.approx(0.5e-3 * (1 + math.cos(math.pi * 15 / 30)))
    opt = cfg.build_optimizer([list(net.get_parameters(["bn", "bias"], mode="exclude")),
                               list(net.get_parameters(["bn", "bias"], mode="include"))])
    assert isinstance(opt, torch.optim.Adam)
    assert [g["weight_decay"] for g in opt.param_groups] == [3e-5, 0]
    assert cfg.warmup_adjust_learning_rate(opt, 20, 10, 0, 4, warmup_lr=0) == pytest.approx(5 / 20 * 1e-3)
    assert "n_epochs" in cfg.config and not any(k.startswith("_") for k in cfg.config)
    dp = cfg.data_provider
    batch = next(iter(dp.train))
    assert set(batch) == {"image", "2x_down_image", "4x_down_image"}
    assert tuple(batch["4x_down_image"].shape[2:]) == (64, 64) and float(batch["image"].max()) <= 1.0


def test_metric_matches_reference(golden, ora):
    u = amd("utils")
    g = golden("metric.npz")
    a, b = torch.from_numpy(g["a"]), torch.from_numpy(g["b"])
    assert np.array_equal(u.tensor2img_np(b), g["u8_b"])
    assert np.array_equal(u.rgb2y(u.tensor2img_np(b)), g["y_b"])
    assert abs(u.psnr_y(a, b) - float(g["psnr_ab"])) < 1e-9
    assert abs(float(u.psnr_y_device(a, b)) - float(g["psnr_ab"])) < 1e-9
    # tensor2img_np must not clamp the caller's tensor (the reference only does on a CPU run)
    assert float(b.min()) < 0.0
    # batch > 1: the device metric equals the host metric over the make_grid mosaic
    x = torch.rand(5, 3, 12, 10, generator=torch.Generator().manual_seed(3))
    y = (x + 0.05 * torch.randn(x.shape, generator=torch.Generator().manual_seed(4))).clamp(-0.2, 1.2)
    assert abs(float(u.psnr_y_device(x, y)) - u.psnr_y(x, y)) < 1e-9
    gs4 = golden("s4_net.npz")
    assert abs(float(u.psnr_y_device(torch.from_numpy(gs4["psnr_y1"]), torch.from_numpy(gs4["psnr_tgt"])))
               - float(gs4["psnr_value"])) < 1e-9


def test_build_activation_and_layers():
    u, L = amd("utils"), amd("layers")
    assert isinstance(u.build_activation("pixelshuffle"), u.PixelShuffle)
    assert isinstance(u.build_activation("pixelunshuffle"), u.PixelUnshuffle)
    assert u.build_activation(None) is None
    seq = u.build_activation("pixelshuffle+relu6")
    assert isinstance(seq[0], u.PixelShuffle) and isinstance(seq[1], torch.nn.ReLU6)
    with pytest.raises(ValueError):
        u.build_activation("h_swish")
    c = L.ConvLayer(64, 256, kernel_size=5, act_func="pixelshuffle", use_bn=True)
    assert list(c._modules) == ["conv", "bn", "act"] and c.module_str == "5x5_Conv_O256"
    assert c.conv.padding == (2, 2) and c.conv.bias is None
    c2 = L.set_layer_from_config(c.config)
    assert isinstance(c2, L.ConvLayer) and c2.config == c.config
    mb = L.MBInvertedConvLayer(64, 64, 5, 1, 4, mid_channels=256)
    assert mb.module_str == "5x5_MBConv4_RELU6_O64"
    assert list(mb.state_dict()) [:1] == ["inverted_bottleneck.conv.weight"]


def test_count_net_flops_matches_survey(net):
    pu = amd("imagenet_codebase.utils")
    net.set_active_subnet(ks=7, e=6, d=4, pixel_d=2)
    type(net).COMPAT_REFERENCE_INDEXING = False
    try:
        net.set_active_subnet(ks=7, e=6, d=4, pixel_d=2)
        macs = pu.count_net_flops(net, (1, 3, 64, 64))
    finally:
        type(net).COMPAT_REFERENCE_INDEXING = True
    # 28.0 GFLOP forward per image for k7/e6/d4 at 64x64 -> 256x256 (= 14.0 GMAC)
    assert abs(2 * macs / 1e9 - 28.0) < 0.3, macs


def test_infer_operand_cache_identity_versions_and_epoch():
    """ops._infer_operands (host logic, CPU tensors): operands are reused only for the SAME tensor objects at the same
    address and version; an in-place update, a new tensor object (even at a recycled address) or clear_infer_cache()
    prepares again; the epoch counter GraphedEval keys on moves with every clear."""
    import gc

    import torch

    ops = amd("ops")
    ops.clear_infer_cache()
    calls = []

    def prepare(ptr, nbytes):
        calls.append(int(nbytes.value))

    w = torch.zeros(8)
    b = torch.ones(8)
    buf0 = ops._infer_operands(("k", 1), (w, b, None), 64, "cpu", prepare)
    buf1 = ops._infer_operands(("k", 1), (w, b, None), 64, "cpu", prepare)
    assert buf0 is buf1 and len(calls) == 1
    assert ops._infer_operands(("k", 2), (w, b, None), 64, "cpu", prepare) is not buf0 and len(calls) == 2   # other kind
    w.add_(1.0)                                                  # tracked in-place write: version bump
    buf2 = ops._infer_operands(("k", 1), (w, b, None), 64, "cpu", prepare)
    assert buf2 is not buf0 and len(calls) == 3
    # a different tensor object never matches an entry made for another one, whatever its address / version
    w2 = w.clone()
    assert ops._infer_operands(("k", 1), (w2, b, None), 64, "cpu", prepare) is not buf2 and len(calls) == 4
    ptr, ver = w2.data_ptr(), w2._version
    del w2
    gc.collect()
    for _ in range(64):                                          # try to get a new tensor at the recycled address
        w3 = torch.empty(8)
        if w3.data_ptr() == ptr and w3._version == ver:
            n = len(calls)
            ops._infer_operands(("k", 1), (w3, b, None), 64, "cpu", prepare)
            assert len(calls) == n + 1, "an entry outlived the tensor it was built from"
            break
    e0 = ops.infer_epoch()
    ops.clear_infer_cache()
    assert ops.infer_epoch() == e0 + 1 and not ops.infer_operand_buffers()
    ops._infer_operands(("k", 1), (w, b, None), 64, "cpu", prepare)
    assert len(ops.infer_operand_buffers()) == 1
    ops.clear_infer_cache()
