"""BASELINE configs 2 and 5 on the GPU, against the network-level CPU oracle (oracle/s4_port.py, pinned to the reference
by tests/golden/s4_net.npz).  `-m gpu`.

C2: S4 2x supernet (pixelshuffle_depth_list=[1]), fixed sub-network k=3 / e=6 / d=4, LR 48x48 -> HR 96x96, N=16, one
    training step: fp32 against the oracle at fp32 tolerances, bf16 (the configuration BASELINE names) against the same
    oracle with the 16-bit realisation bounds measured on MI355X.
C5: eval_ofa_net_sr.py's path -- sub-network (ks=7, e=6, d=2, pixel_d=2), eval-mode BN, fp32, Set14-like LR sizes with
    odd widths (120x125, 90x62, 97x146): |dPSNR| <= 1e-3 dB through the reference's metric (north_star), and the
    size-bucketed batched validation equals the batch-1 pass (reference eval_ofa_net_sr.py:187-220,247-251,
    div2k_setxx.py:182-190, sr_run_manager.py:323-393)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, amd, assert_close
from detfill import det_uniform, fill_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _port_sd(net):
    from oracle import s4_port
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
    return sd


def _rel(a, b):
    return float((a.detach().double().cpu() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.fixture(scope="module")
def nets():
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    return amd("elastic_nn.networks")


def test_config2_train_step_fp32_and_bf16(nets):
    from oracle import s4_port
    torch.manual_seed(5)
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                              pixelshuffle_depth_list=[1])
    net.init_model("he_fout")
    net.to(DEV).train()
    net.set_active_subnet(ks=3, e=6, d=4, pixel_d=1)
    assert net.active_upscale() == 2
    arch = s4_port.Arch(pd_list=(1,))
    arch.set_active_subnet(ks=3, e=6, d=4, pixel_d=1)
    g = torch.Generator().manual_seed(3)
    hr = torch.rand((16, 3, 96, 96), generator=g)
    lr = F.interpolate(hr, scale_factor=0.5, mode="bicubic", antialias=True).clamp_(0, 1)
    sd = _port_sd(net)
    y_ref = s4_port.s4_forward(sd, lr, arch, training=True)
    loss_ref = F.mse_loss(y_ref, hr)
    loss_ref.backward()
    sd_gpu0 = {k: v.detach().clone() for k, v in net.state_dict().items()}

    def step(dtype):
        net.load_state_dict(sd_gpu0)
        net.zero_grad(set_to_none=True)
        if dtype is None:
            y = net(lr.to(DEV))
        else:
            with torch.autocast("cuda", dtype=dtype):
                y = net(lr.to(DEV))
        loss = F.mse_loss(y.float(), hr.to(DEV))
        loss.backward()
        amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
        ga = torch.cat([p.grad.detach().double().flatten().cpu() for n, p in net.named_parameters() if p.grad is not None])
        gb = torch.cat([sd[n].grad.double().flatten() for n, p in net.named_parameters() if p.grad is not None])
        for n, p in net.named_parameters():
            assert (p.grad is None) == (sd[n].grad is None), n
        return y, float(loss), float((ga - gb).norm() / gb.norm())

    y, loss, gerr = step(None)
    assert tuple(y.shape) == (16, 3, 96, 96)
    assert abs(loss - float(loss_ref)) <= 2e-5 * float(loss_ref)
    assert_close(y.detach().cpu().numpy(), y_ref.detach().numpy(), 1e-3, 1e-4, "C2 fp32 output")
    assert gerr <= 2e-3, gerr                        # whole-net relative gradient error, fp32
    for n in ("running_mean", "running_var"):
        k = "blocks.15.mobile_inverted_conv.depth_conv.bn.bn." + n
        assert_close(net.state_dict()[k].cpu().numpy(), sd[k].numpy(), 1e-4, 1e-5, k)
    yb, lossb, gerrb = step(torch.bfloat16)
    print("C2 bf16: loss %.6f vs %.6f, output rel %.4g, whole-net gradient rel %.4g" % (
        lossb, float(loss_ref), _rel(yb.float(), y_ref.detach()), gerrb))
    assert yb.dtype == torch.bfloat16
    assert abs(lossb - float(loss_ref)) <= 1e-2 * float(loss_ref)
    assert _rel(yb.float(), y_ref.detach()) <= 5e-2       # ~60 layers of 16-bit storage: 2.8 % measured on MI355X
    assert gerrb <= 0.2, gerrb


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_config3_full_size_step_vs_oracle(dtype):
    """BASELINE config 3 at the size bench.py times it -- S4 4x supernet, elastic kernel {3,5,7}, N=16, LR 64x64 -> HR
    256x256, the objects of bench.TrainWorkload -- against oracle/s4_port.py on the host cores, sub-network seeds 0 and 1
    (reference progressive_shrinking.py:152-203): loss of the step, None mask of every parameter gradient, whole-net
    relative gradient error; fp32 at fp32 tolerance, bf16 at the realisation bound measured on MI355X (DESIGN.md 4:
    a 16-bit training step carries per-stage parity bars -- tests/test_hip_composite16.py -- and only this bound end to
    end)."""
    import random

    import bench
    from oracle import s4_port
    dev = torch.device("cuda", 0)
    wl = bench.TrainWorkload(bench.mods(), "c3", dev, batch=16, lr_size=64, dtype=dtype)
    arch = s4_port.Arch(ks_list=(3, 5, 7), expand_list=(6,), depth_list=(4,), pd_list=(2,))
    hr, lr = wl.hr.cpu(), wl.lr.cpu()
    assert tuple(lr.shape) == (16, 3, 64, 64) and tuple(hr.shape) == (16, 3, 256, 256)
    for i in (0, 1):
        sd = _port_sd(wl.net)                      # the weights this step starts from (step 1: after the GPU's Adam step)
        random.seed(bench.subnet_seed(i))
        arch.sample_active_subnet()
        loss_ref = F.mse_loss(s4_port.s4_forward(sd, lr, arch, training=True), hr)
        loss_ref.backward()
        loss = float(wl.step(i))                   # forward + backward + Adam on the GPU, same seed rule
        named = list(wl.net.named_parameters())
        for n, p in named:
            assert (p.grad is None) == (sd[n].grad is None), (i, n)
        ga = torch.cat([p.grad.detach().double().flatten().cpu() for n, p in named if p.grad is not None])
        gb = torch.cat([sd[n].grad.double().flatten() for n, p in named if p.grad is not None])
        gerr = float((ga - gb).norm() / gb.norm())
        print("C3 full size %s seed %d: loss %.6f vs %.6f, whole-net gradient rel %.4g" % (dtype, i, loss, float(loss_ref), gerr))
        if dtype == "f32":
            assert abs(loss - float(loss_ref)) <= 2e-5 * float(loss_ref)
            assert gerr <= 2e-3, gerr
        else:
            assert abs(loss - float(loss_ref)) <= 1e-2 * float(loss_ref)
            assert gerr <= 0.2, gerr


SIZES = [(120, 125), (90, 62), (97, 146)]


def test_config5_eval_psnr_parity_at_set14_like_sizes(nets, ora):
    from oracle import s4_port
    meta = json.load(open(os.path.join(GOLDEN, "s4_meta.json")))
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                              pixelshuffle_depth_list=[1, 2])
    shapes = {k: tuple(v) for k, v in meta["state_dict_shapes"].items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(shapes, "s4").items()})
    net.to(DEV).eval()
    net.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)                 # eval_ofa_net_sr.py:207-220
    arch = s4_port.Arch()
    arch.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    for (h, w) in SIZES:
        lr = torch.from_numpy(det_uniform((1, 3, h, w), "c5/lr%dx%d" % (h, w), 0.0, 1.0))
        with torch.no_grad():
            y_ref = s4_port.s4_forward(sd, lr, arch, training=False)
            y = net(lr.to(DEV))
        assert tuple(y.shape) == (1, 3, 4 * h, 4 * w)
        assert_close(y.cpu().numpy(), y_ref.numpy(), 1e-3, 1e-4, "C5 output %dx%d" % (h, w))
        noise = torch.from_numpy(det_uniform(tuple(y_ref.shape), "c5/hr%dx%d" % (h, w), 0.0, 1.0))
        tgt = (0.7 * y_ref.clamp(0, 1) + 0.3 * noise).clamp(0, 1)
        p_ref = ora.psnr_y(y_ref.numpy(), tgt.numpy())
        p_gpu = amd("utils").psnr_y(y, tgt)
        assert abs(p_gpu - p_ref) <= 1e-3, ((h, w), p_gpu, p_ref)


def test_config5_batched_validation_equals_batch1(nets, tmp_path):
    rm = amd("imagenet_codebase.run_manager")
    torch.manual_seed(9)
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                              pixelshuffle_depth_list=[1, 2])
    sizes = [(288, 352), (248, 360), (256, 256), (288, 352), (256, 256), (256, 256), (276, 276)]
    cfg = rm.SyntheticSRRunConfig(n_epochs=1, init_lr=1e-3, train_batch_size=1, test_batch_size=1, image_size=64,
                                  n_train_batches=1, test_sizes=sizes)
    args = argparse.Namespace(ks_list=[3, 5, 7], expand_list=[3, 4, 6], depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    mgr = rm.SRRunManager(str(tmp_path), net, cfg, init=True, num_gpus=1, args=args)
    with torch.no_grad():     # non-trivial BN statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.7, 1.3)
    net.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)
    l1, p1 = mgr.validate(is_test=True, input_key="4x_down_image")
    lb, pb, calls = mgr.validate_batched(is_test=True, input_key="4x_down_image")
    assert calls == 4                                  # 7 images, 4 distinct sizes
    assert abs(lb - l1) <= 1e-5 * abs(l1) and abs(pb - p1) <= 1e-3, ((l1, p1), (lb, pb))
    lc, pc, calls2 = mgr.validate_batched(is_test=True, input_key="4x_down_image", max_batch=2)
    assert calls2 == 5 and abs(lc - l1) <= 1e-5 * abs(l1) and abs(pc - p1) <= 1e-3
    utils = amd("utils")
    groups = utils.bucket_by_size([torch.zeros(1, 3, h, w) for h, w in sizes])
    assert [len(g_) for g_ in groups] == [2, 1, 3, 1] and all(len({tuple(t.shape) for t in g_}) == 1 for g_ in groups)
