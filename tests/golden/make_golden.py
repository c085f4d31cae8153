#!/usr/bin/env python3
"""Generate golden fixtures by running the REFERENCE (imported from /root/reference) on seeded
inputs.  Runs only in the build container; the reference never travels.  Output: small .npz /
.json files next to this script holding plain arrays (inputs where they are not re-derivable
from tests/golden/detfill.py, and expected outputs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

torchvision is not installed here; the reference imports it at module level only for an
image-dump helper (make_grid) and for dataset plumbing, so an inert stub is registered in
sys.modules *inside this script only* (SURVEY.md section 8c).
"""
import json
import math
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OFASR_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from detfill import det_uniform, det_ints, fill_state_dict  # noqa: E402


def _stub_torchvision():
    def make_grid(tensor, nrow=8, padding=2, normalize=False, **kw):
        # batch-1 behaviour of torchvision.utils.make_grid: squeeze.  Larger batches are not
        # used by the goldens.
        if tensor.dim() == 4 and tensor.size(0) == 1:
            return tensor.squeeze(0)
        # torchvision.utils.make_grid restated for batches > 1 (torchvision itself is absent; this layout is
        # torchvision's documented one, NOT the reference's code -- "parity unpinned" for the mosaic, see gen_teacher):
        # xmaps = min(nrow, N) columns, cells of (H + padding) x (W + padding), zero fill, image (y, x) at
        # (y*(H+padding) + padding, x*(W+padding) + padding)
        n, c, h, w = tensor.shape
        xmaps = min(nrow, n)
        ymaps = int(math.ceil(float(n) / xmaps))
        ch, cw = h + padding, w + padding
        grid = tensor.new_zeros((c, ch * ymaps + padding, cw * xmaps + padding))
        k = 0
        for yy in range(ymaps):
            for xx in range(xmaps):
                if k >= n:
                    break
                grid[:, yy * ch + padding:yy * ch + padding + h, xx * cw + padding:xx * cw + padding + w] = tensor[k]
                k += 1
        return grid

    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = make_grid
    tvt = types.ModuleType("torchvision.transforms")
    tvtf = types.ModuleType("torchvision.transforms.functional")
    tvd = types.ModuleType("torchvision.datasets")
    tvm = types.ModuleType("torchvision.models")

    class _Placeholder(object):
        def __init__(self, *a, **k):
            pass

    for name in ["RandomResizedCrop", "Compose", "ToTensor", "Normalize", "RandomHorizontalFlip",
                 "Resize", "CenterCrop", "ColorJitter", "RandomCrop", "Lambda"]:
        setattr(tvt, name, type(name, (_Placeholder,), {}))
    tvm.Inception3 = type("Inception3", (_Placeholder,), {})
    tvd.ImageFolder = type("ImageFolder", (_Placeholder,), {})
    tv.utils, tv.transforms, tv.datasets, tv.models = tvu, tvt, tvd, tvm
    tvt.functional = tvtf
    for k, m in [("torchvision", tv), ("torchvision.utils", tvu), ("torchvision.transforms", tvt),
                 ("torchvision.transforms.functional", tvtf), ("torchvision.datasets", tvd),
                 ("torchvision.models", tvm)]:
        sys.modules[k] = m


_stub_torchvision()

from ofa.elastic_nn.modules.dynamic_op import (  # noqa: E402
    DynamicSeparableConv2d, DynamicPointConv2d, DynamicBatchNorm2d)
from ofa.elastic_nn.modules.dynamic_layers import DynamicMBConvLayer  # noqa: E402
from ofa.utils import pixel_unshuffle, psnr as ref_psnr  # noqa: E402

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def A(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name if name.endswith(".npz") else name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %8.1f KB" % (name, os.path.getsize(path) / 1024.0))


# ----------------------------------------------------------------------------- a1 pointwise
def gen_pwconv():
    out = {}
    # expand: weight [384,64,1,1], active out in {192,256,384}; ragged HW (6x7=42)
    conv = DynamicPointConv2d(64, 384)
    w = det_uniform((384, 64, 1, 1), "pw/expand/w", -0.2, 0.2)
    conv.conv.weight.data.copy_(T(w))
    x = det_uniform((2, 64, 6, 7), "pw/expand/x")
    out["expand_w"] = w
    out["expand_x"] = x
    for oc in (192, 256, 384):
        xt = T(x).requires_grad_(True)
        conv.zero_grad()
        y = conv(xt, oc)
        dy = det_uniform(tuple(y.shape), "pw/expand/dy%d" % oc)
        y.backward(T(dy))
        out["expand_y_%d" % oc] = A(y)
        out["expand_dx_%d" % oc] = A(xt.grad)
        out["expand_dw_%d" % oc] = A(conv.conv.weight.grad)
    # project: weight [64,384,1,1], in-channels = x.size(1) in {192,256,384} (strided row slice)
    conv = DynamicPointConv2d(384, 64)
    w = det_uniform((64, 384, 1, 1), "pw/project/w", -0.1, 0.1)
    conv.conv.weight.data.copy_(T(w))
    out["project_w"] = w
    for ic in (192, 256, 384):
        x = det_uniform((2, ic, 6, 7), "pw/project/x%d" % ic)
        xt = T(x).requires_grad_(True)
        conv.zero_grad()
        y = conv(xt)
        dy = det_uniform(tuple(y.shape), "pw/project/dy%d" % ic)
        y.backward(T(dy))
        out["project_y_%d" % ic] = A(y)
        out["project_dx_%d" % ic] = A(xt.grad)
        out["project_dw_%d" % ic] = A(conv.conv.weight.grad)
    save("pwconv.npz", **out)


# --------------------------------------------------------------------- a2/a3 depthwise + transform
def gen_dwconv():
    out = {}
    CMAX, H, W = 24, 9, 11
    w7 = det_uniform((CMAX, 1, 7, 7), "dw/w7", -0.3, 0.3)
    m75 = (np.eye(25, dtype=np.float32) + det_uniform((25, 25), "dw/m75", -0.2, 0.2)).astype(np.float32)
    m53 = (np.eye(9, dtype=np.float32) + det_uniform((9, 9), "dw/m53", -0.2, 0.2)).astype(np.float32)
    out.update(w7=w7, m75=m75, m53=m53)
    for mode in (None, 1):
        DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = mode
        op = DynamicSeparableConv2d(CMAX, [3, 5, 7])
        op.conv.weight.data.copy_(T(w7))
        if mode is not None:
            getattr(op, "7to5_matrix").data.copy_(T(m75))
            getattr(op, "5to3_matrix").data.copy_(T(m53))
        for C in (16, 24):
            x = det_uniform((2, C, H, W), "dw/x%d" % C)
            out["x_%d" % C] = x
            for k in (3, 5, 7):
                tag = "m%s_c%d_k%d" % ("N" if mode is None else "1", C, k)
                op.zero_grad()
                xt = T(x).requires_grad_(True)
                filt = op.get_active_filter(C, k).contiguous()
                y = op(xt, k)
                dy = det_uniform(tuple(y.shape), "dw/dy/" + tag)
                y.backward(T(dy))
                out["filter_" + tag] = A(filt)
                out["y_" + tag] = A(y)
                out["dx_" + tag] = A(xt.grad)
                out["dw7_" + tag] = A(op.conv.weight.grad)
                if mode is not None:
                    g75 = getattr(op, "7to5_matrix").grad
                    g53 = getattr(op, "5to3_matrix").grad
                    out["dm75_isnone_" + tag] = np.array(g75 is None)
                    out["dm53_isnone_" + tag] = np.array(g53 is None)
                    if g75 is not None:
                        out["dm75_" + tag] = A(g75)
                    if g53 is not None:
                        out["dm53_" + tag] = A(g53)
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("dwconv.npz", **out)


# ------------------------------------------------------------------------------------- a4 BN
def gen_bn():
    out = {}
    CMAX = 24
    for C in (16, 24):
        for training in (True, False):
            bn = DynamicBatchNorm2d(CMAX)
            bn.bn.momentum, bn.bn.eps = 0.1, 1e-5
            sd = fill_state_dict({"bn.weight": (CMAX,), "bn.bias": (CMAX,),
                                  "bn.running_mean": (CMAX,), "bn.running_var": (CMAX,)}, "bnfix")
            bn.bn.weight.data.copy_(T(sd["bn.weight"]))
            bn.bn.bias.data.copy_(T(sd["bn.bias"]))
            bn.bn.running_mean.copy_(T(sd["bn.running_mean"]))
            bn.bn.running_var.copy_(T(sd["bn.running_var"]))
            bn.train(training)
            x = det_uniform((3, C, 5, 6), "bn/x%d" % C, -2.0, 2.0)
            xt = T(x).requires_grad_(True)
            y = bn(xt)
            dy = det_uniform(tuple(y.shape), "bn/dy%d" % C)
            y.backward(T(dy))
            tag = "c%d_%s" % (C, "train" if training else "eval")
            out["y_" + tag] = A(y)
            out["dx_" + tag] = A(xt.grad)
            out["dgamma_" + tag] = A(bn.bn.weight.grad)
            out["dbeta_" + tag] = A(bn.bn.bias.grad)
            out["rm_" + tag] = A(bn.bn.running_mean)
            out["rv_" + tag] = A(bn.bn.running_var)
            out["nbt_" + tag] = A(bn.bn.num_batches_tracked)
    save("bn.npz", **out)


# ----------------------------------------------------------------------------- a8/a9 shuffle
def gen_pixelshuffle():
    out = {}
    for (N, C, H, W) in [(2, 3, 5, 7), (1, 16, 4, 4)]:
        x = det_ints((N, C * 4, H, W), "ps/x%d_%d" % (C, H), -64, 64)
        y = nn.PixelShuffle(2)(T(x))
        tag = "%d_%d_%d_%d" % (N, C, H, W)
        out["shuffle_y_" + tag] = A(y)
        # reference PixelUnshuffle (one-hot strided conv) on an integer tensor
        z = det_ints((N, C, H * 2, W * 2), "pus/x%d_%d" % (C, H), -64, 64)
        u = pixel_unshuffle(T(z), 2)
        out["unshuffle_y_" + tag] = A(u)
    save("pixelshuffle.npz", **out)


# ------------------------------------------------------------------------------ a5/a6 block
def _load_sd(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = fill_state_dict(shapes, prefix)
    module.load_state_dict({k: T(v) for k, v in sd.items()})
    return sd


def gen_mbblock():
    from ofa.imagenet_codebase.networks.proxyless_nets import MobileInvertedResidualBlock
    from ofa.layers import IdentityLayer
    out = {}
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    C = 16
    layer = DynamicMBConvLayer([C], [C], [3, 5, 7], [3, 4, 6], stride=1, act_func="relu6")
    block = MobileInvertedResidualBlock(layer, IdentityLayer([C], [C]))
    for m in block.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum, m.eps = 0.1, 1e-5
    x = det_uniform((2, C, 10, 9), "mb/x")
    out["x"] = x
    for bn_train in (True, False):
        for (k, e) in [(7, 6), (5, 4), (3, 3), (3, 6), (7, 3)]:
            _load_sd(block, "mbblock")
            block.train(bn_train)
            layer.active_kernel_size, layer.active_expand_ratio = k, e
            block.zero_grad()
            xt = T(x).requires_grad_(True)
            y = block(xt)
            dy = det_uniform(tuple(y.shape), "mb/dy")
            y.backward(T(dy))
            tag = "k%d_e%d_%s" % (k, e, "train" if bn_train else "eval")
            out["y_" + tag] = A(y)
            out["dx_" + tag] = A(xt.grad)
            for name, p in block.named_parameters():
                out["isnone_%s_%s" % (name, tag)] = np.array(p.grad is None)
                if p.grad is not None:
                    out["grad_%s_%s" % (name, tag)] = A(p.grad)
            if bn_train:
                for name, b in block.named_buffers():
                    out["buf_%s_%s" % (name, tag)] = A(b)
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("mbblock.npz", **out)


# --------------------------------------------------------------------------------- a10 network
def _grad_summary(net):
    names, isnone, s1, sabs, l2 = [], [], [], [], []
    for name, p in net.named_parameters():
        names.append(name)
        if p.grad is None:
            isnone.append(True)
            s1.append(0.0)
            sabs.append(0.0)
            l2.append(0.0)
        else:
            g = p.grad.double()
            isnone.append(False)
            s1.append(float(g.sum()))
            sabs.append(float(g.abs().sum()))
            l2.append(float(g.pow(2).sum().sqrt()))
    return names, np.array(isnone), np.array(s1), np.array(sabs), np.array(l2)


FULL_GRAD_KEYS = [
    "dec_first_conv_block.conv.weight",
    "blocks.0.mobile_inverted_conv.inverted_bottleneck.conv.conv.weight",
    "blocks.0.mobile_inverted_conv.depth_conv.conv.conv.weight",
    "blocks.0.mobile_inverted_conv.depth_conv.conv.7to5_matrix",
    "blocks.0.mobile_inverted_conv.depth_conv.conv.5to3_matrix",
    "blocks.5.mobile_inverted_conv.point_linear.conv.conv.weight",
    "blocks.5.mobile_inverted_conv.depth_conv.bn.bn.weight",
    "blocks.16.bn.bias",
    "dec_final_output_conv_block.conv.weight",
]


def gen_s4():
    from ofa.elastic_nn.networks import OFAMobileNetS4
    out = {}
    meta = {}
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                         pixelshuffle_depth_list=[1, 2])
    meta["state_dict_shapes"] = {k: list(v.shape) for k, v in net.state_dict().items()}
    meta["param_names"] = [n for n, _ in net.named_parameters()]
    meta["block_group_info"] = net.block_group_info
    meta["n_params"] = int(sum(p.numel() for p in net.parameters()))
    # optimizer grouping facts (sr_run_manager.py:180-191)
    keys = ["bn", "bias"]
    meta["n_decay"] = len(list(net.get_parameters(keys, mode="exclude")))
    meta["n_no_decay"] = len(list(net.get_parameters(keys, mode="include")))

    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, "s4")

    lr = det_uniform((2, 3, 12, 10), "s4/lr", 0.0, 1.0)
    out["lr"] = lr
    settings = [
        dict(ks=7, e=6, d=4, pixel_d=2),
        dict(ks=3, e=3, d=2, pixel_d=2),
        dict(ks=5, e=4, d=3, pixel_d=1),
    ]
    meta["settings"] = settings
    meta["runtime_depth"] = []
    meta["out_shapes"] = []
    for si, setting in enumerate(settings):
        for bn_train in (True, False):
            net.load_state_dict({k: T(v) for k, v in sd.items()})
            net.train(bn_train)
            net.set_active_subnet(**setting)
            if bn_train:
                meta["runtime_depth"].append(list(net.runtime_depth))
            net.zero_grad()
            y = net(T(lr))
            hr = det_uniform(tuple(y.shape), "s4/hr%d" % si, 0.0, 1.0)
            loss = F.mse_loss(y, T(hr))
            loss.backward()
            tag = "s%d_%s" % (si, "train" if bn_train else "eval")
            if bn_train:
                meta["out_shapes"].append(list(y.shape))
            out["y_" + tag] = A(y)
            out["loss_" + tag] = np.array(float(loss))
            names, isnone, s1, sabs, l2 = _grad_summary(net)
            out["g_isnone_" + tag] = isnone
            out["g_sum_" + tag] = s1
            out["g_abs_" + tag] = sabs
            out["g_l2_" + tag] = l2
            gd = dict(net.named_parameters())
            for k in FULL_GRAD_KEYS:
                if gd[k].grad is not None:
                    out["grad_%s_%s" % (k, tag)] = A(gd[k].grad)
            if bn_train:
                bufs = dict(net.named_buffers())
                for k in ["blocks.0.mobile_inverted_conv.depth_conv.bn.bn.running_mean",
                          "blocks.0.mobile_inverted_conv.depth_conv.bn.bn.running_var",
                          "blocks.0.mobile_inverted_conv.depth_conv.bn.bn.num_batches_tracked",
                          "blocks.3.mobile_inverted_conv.depth_conv.bn.bn.num_batches_tracked",
                          "blocks.16.bn.running_mean"]:
                    out["buf_%s_%s" % (k, tag)] = A(bufs[k])
    # PSNR of a batch-1 forward through the reference's own metric code path
    from ofa.elastic_nn.training.progressive_shrinking import tensor2img_np, rgb2y
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    net.eval()
    net.set_active_subnet(ks=7, e=6, d=4, pixel_d=2)
    with torch.no_grad():
        y1 = net(T(lr[:1]))
    hr1 = det_uniform(tuple(y1.shape), "s4/hr_psnr", 0.0, 1.0)
    # make the target close to the output so the PSNR is in a realistic range
    tgt = (0.7 * y1.clamp(0, 1) + 0.3 * T(hr1)).clamp(0, 1)
    out["psnr_y1"] = A(y1)
    out["psnr_tgt"] = A(tgt)
    out["psnr_value"] = np.array(ref_psnr(rgb2y(tensor2img_np(y1.clone())), rgb2y(tensor2img_np(tgt.clone()))))

    # sampling traces (progressive_shrinking.py:164-165 seed rule; ofa_mbs4.py:316-370)
    traces = []
    for step in (0, 1, 7, 123):
        for sub in (0, 1):
            seed = int('%d%.3d%.3d' % (step, sub, 0))
            random.seed(seed)
            s = net.sample_active_subnet()
            traces.append(dict(seed=seed, sampled={k: v for k, v in s.items()},
                               runtime_depth=list(net.runtime_depth),
                               ks=[b.mobile_inverted_conv.active_kernel_size for b in net.blocks[:-2]],
                               e=[b.mobile_inverted_conv.active_expand_ratio for b in net.blocks[:-2]]))
    meta["sample_traces"] = traces
    # constrained sampling (set_constraint) trace
    net.set_constraint([4, 3], constraint_type="depth")
    net.set_constraint([7, 5], constraint_type="kernel_size")
    random.seed(2021)
    s = net.sample_active_subnet()
    meta["constrained_trace"] = dict(seed=2021, sampled=s, runtime_depth=list(net.runtime_depth))
    net.clear_constraint()
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("s4_net.npz", **out)
    with open(os.path.join(HERE, "s4_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote s4_meta.json")


# ------------------------------------------------------------------------- metric on an image
def gen_metric():
    from PIL import Image
    from ofa.elastic_nn.training.progressive_shrinking import tensor2img_np, rgb2y
    img = np.asarray(Image.open(os.path.join(REF, "zssr.png")).convert("RGB"))
    crop = img[40:72, 50:90, :].astype(np.float32) / 255.0  # 32x40 natural-image crop
    a = np.transpose(crop, (2, 0, 1))[None]
    noise = det_uniform(a.shape, "metric/noise", -0.08, 0.08)
    b = a + noise  # leaves [0,1] in places: exercises the clamp
    out = dict(a=a.astype(np.float32), b=b.astype(np.float32))
    out["u8_b"] = tensor2img_np(T(b).clone())
    out["y_b"] = rgb2y(tensor2img_np(T(b).clone()))
    out["psnr_ab"] = np.array(ref_psnr(rgb2y(tensor2img_np(T(a).clone())), rgb2y(tensor2img_np(T(b).clone()))))
    save("metric.npz", **out)


# ------------------------------------------------------------------ a11 progressive-shrinking steps
def gen_calibration():
    """BN re-calibration of a sampled sub-network (reference ofa/elastic_nn/utils.py:16-64) on two batches of
    different size (pins the batch-size weighting of the reference's AverageMeter)."""
    from ofa.elastic_nn.networks import OFAMobileNetS4
    from ofa.elastic_nn.utils import set_running_statistics
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                         pixelshuffle_depth_list=[1, 2])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, "s4")
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    net.eval()
    net.set_active_subnet(ks=5, e=4, d=3, pixel_d=2)
    b0 = det_uniform((2, 3, 12, 10), "cal/b0", 0.0, 1.0)
    b1 = det_uniform((3, 3, 12, 10), "cal/b1", 0.0, 1.0)
    loader = [{"image": T(b0)}, {"image": T(b1)}]
    set_running_statistics(net, loader)
    out = {"b0": b0, "b1": b1}
    for k, v in net.state_dict().items():
        if "running_mean" in k or "running_var" in k:
            out[k] = A(v)
    save("calibration", **out)


def gen_div2k():
    """deterministic pieces of the Div2K/SetXX data path, through the reference's own PIL code
    (ofa/imagenet_codebase/data_providers/div2k_setxx.py:246-379): ModCrop(4), Scale(1/2), Scale(1/4) of one image and
    the (duplicating) recursive file listing."""
    import tempfile
    from PIL import Image
    from ofa.imagenet_codebase.data_providers.div2k_setxx import ModCrop, Scale, get_image_paths_recursive
    from detfill import det_ints
    hr = det_ints((37, 50, 3), "div2k/hr", 0, 256).astype(np.uint8)
    img = Image.fromarray(hr, "RGB")
    H = ModCrop(mod=4)(img)
    L2 = Scale(scale_factor=1 / 2)(H)
    L4 = Scale(scale_factor=1 / 4)(H)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "sub", "deeper"))
        for rel in ("a.png", "z.txt", os.path.join("sub", "b.png"), os.path.join("sub", "deeper", "c.jpg")):
            open(os.path.join(d, rel), "wb").close()
        listing = [os.path.relpath(q, d) for q in get_image_paths_recursive(d, [])]
    save("div2k", hr=hr, H=np.asarray(H), L2=np.asarray(L2), L4=np.asarray(L4),
         listing=np.array("|".join(listing)))


def gen_trainer():
    """two optimizer steps of the progressive-shrinking hot loop (reference progressive_shrinking.py:152-203,
    transcribed around the REFERENCE net/optimizer because the original hard-codes .cuda()): per step,
    dynamic_batch_size=2 sub-networks sampled with the reference's seed rule, MSE loss, grads accumulated,
    Adam (weight-decay groups of sr_run_manager.py:180-191) stepped once."""
    from ofa.elastic_nn.networks import OFAMobileNetS4
    from ofa.imagenet_codebase.run_manager.sr_run_manager import RunConfig
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                         pixelshuffle_depth_list=[2])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, "s4")
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    net.train()
    cfg = RunConfig(n_epochs=1, init_lr=1e-3, lr_schedule_type="cosine", lr_schedule_param=None, dataset="x",
                    train_batch_size=2, test_batch_size=1, valid_size=None, opt_type="adam", opt_param=None,
                    weight_decay=3e-5, label_smoothing=0.0, no_decay_keys="bn#bias", mixup_alpha=None,
                    model_init="he_fout", validation_frequency=1, print_frequency=1)
    keys = cfg.no_decay_keys.split("#")
    opt = cfg.build_optimizer([net.get_parameters(keys, mode="exclude"), net.get_parameters(keys, mode="include")])
    nBatch, epoch = 2, 0
    out = {}
    hr = det_uniform((2, 2, 3, 32, 32), "tr/hr", 0.0, 1.0)
    x4 = det_uniform((2, 2, 3, 8, 8), "tr/x4", 0.0, 1.0)
    out["hr"], out["x4"] = hr, x4
    losses, lrs, sampled = [], [], []
    for i in range(nBatch):
        lrs.append(cfg.adjust_learning_rate(opt, epoch, i, nBatch))
        opt.zero_grad()
        for sub in range(2):
            random.seed(int('%d%.3d%.3d' % (epoch * nBatch + i, sub, 0)))
            s = net.sample_active_subnet()
            sampled.append({k: v for k, v in s.items()})
            y = net(T(x4[i]))
            loss = F.mse_loss(y, T(hr[i]))
            losses.append(float(loss.detach()))
            loss.backward()
        opt.step()
    out["losses"] = np.array(losses)
    out["lrs"] = np.array(lrs)
    names = [n for n, _ in net.named_parameters()]
    out["w_sum"] = np.array([float(p.detach().double().sum()) for _, p in net.named_parameters()])
    out["w_l2"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for _, p in net.named_parameters()])
    sd0 = fill_state_dict(shapes, "s4")
    out["w_changed"] = np.array([not np.array_equal(sd0[n], A(p)) for n, p in net.named_parameters()])
    pd = dict(net.named_parameters())
    for k in FULL_GRAD_KEYS:
        out["w_" + k] = A(pd[k])
    bufs = dict(net.named_buffers())
    for k in ["blocks.0.mobile_inverted_conv.depth_conv.bn.bn.running_mean",
              "blocks.0.mobile_inverted_conv.depth_conv.bn.bn.num_batches_tracked",
              "blocks.15.mobile_inverted_conv.depth_conv.bn.bn.num_batches_tracked",
              "dec_final_output_conv_block.bn.running_var"]:
        out["buf_" + k] = A(bufs[k])
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("trainer.npz", **out)
    with open(os.path.join(HERE, "trainer_meta.json"), "w") as f:
        json.dump({"param_names": names, "sampled": sampled}, f, indent=1)
    print("wrote trainer_meta.json")


def gen_teacher():
    """the fixed-architecture ("teacher") loop through the REFERENCE's own SRRunManager on CPU
    (sr_run_manager.py:138-198 ctor, :413-514 train_one_epoch with frozen BN, :323-393 validate): S4(ks=[5], e=[3],
    d=[2], pd=[1]) as train_teacher_net_sr_simple.py:186 builds it, Adam with the bn/bias no-decay groups, cosine LR,
    three fixed batches [4,3,32,32] / [4,3,16,16], then validate() on two batch-1 images of different size.
    Recorded: (loss, psnr) of the epoch and of the validation, every post-epoch parameter, the BN buffers (untouched:
    frozen).  The training PSNR goes through make_grid (batch 4 mosaic) -- restated above from
    torchvision's documented layout because torchvision is absent; the batch-1 validation PSNR does not."""
    import argparse
    import tempfile
    from ofa.elastic_nn.networks import OFAMobileNetS4
    from ofa.imagenet_codebase.run_manager.sr_run_manager import RunConfig, SRRunManager
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = OFAMobileNetS4(ks_list=[5], expand_ratio_list=[3], depth_list=[2], pixelshuffle_depth_list=[1])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}

    class Provider(object):
        data_shape = (3, 32, 32)
        image_size = 32

        def __init__(self):
            self.train = [{"image": T(det_uniform((4, 3, 32, 32), "teach/hr%d" % i, 0.0, 1.0)),
                           "2x_down_image": T(det_uniform((4, 3, 16, 16), "teach/x2_%d" % i, 0.0, 1.0))} for i in range(3)]
            self.test = [{"image": T(det_uniform((1, 3) + hw, "teach/vhr%d" % i, 0.0, 1.0)),
                          "2x_down_image": T(det_uniform((1, 3, hw[0] // 2, hw[1] // 2), "teach/vx2_%d" % i, 0.0, 1.0))}
                         for i, hw in enumerate([(32, 32), (24, 40)])]
            self.valid = self.test

    class Cfg(RunConfig):
        _prov = Provider()

        @property
        def data_provider(self):
            return self._prov

    cfg = Cfg(n_epochs=2, init_lr=1e-4, lr_schedule_type="cosine", lr_schedule_param=None, dataset="x",
              train_batch_size=4, test_batch_size=1, valid_size=None, opt_type="adam", opt_param=None,
              weight_decay=3e-5, label_smoothing=0.0, no_decay_keys="bn#bias", mixup_alpha=None,
              model_init="he_fout", validation_frequency=1, print_frequency=1)
    # the args the script hands over (train_teacher_net_sr_simple.py:79-126, lists as :165-180 converts them)
    args = argparse.Namespace(teacher_model=None, kd_ratio=0.0, kd_type=None, ks_list=[5], expand_list=[3],
                              depth_list=[2], pixelshuffle_depth_list=[1])
    with tempfile.TemporaryDirectory() as d:
        mgr = SRRunManager(d, net, cfg, init=True, no_gpu=True, num_gpus=1, args=args)
        sd = fill_state_dict(shapes, "teach")
        net.load_state_dict({k: T(v) for k, v in sd.items()})
        out = {}
        v0 = mgr.validate(is_test=True, no_logs=True)
        out["valid_before"] = np.array(v0, dtype=np.float64)
        tr = mgr.train_one_epoch(args, 0)
        out["train_epoch0"] = np.array(tr, dtype=np.float64)
        v1 = mgr.validate(is_test=True, no_logs=True)
        out["valid_after"] = np.array(v1, dtype=np.float64)
    names = [n for n, _ in net.named_parameters()]
    out["w_sum"] = np.array([float(p.detach().double().sum()) for _, p in net.named_parameters()])
    out["w_l2"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for _, p in net.named_parameters()])
    out["w_changed"] = np.array([not np.array_equal(sd[n], A(p)) for n, p in net.named_parameters()])
    pd = dict(net.named_parameters())
    for k in ["dec_first_conv_block.conv.weight", "blocks.0.mobile_inverted_conv.inverted_bottleneck.conv.conv.weight",
              "blocks.0.mobile_inverted_conv.depth_conv.conv.conv.weight",
              "blocks.7.mobile_inverted_conv.point_linear.conv.conv.weight",
              "blocks.7.mobile_inverted_conv.depth_conv.bn.bn.weight", "dec_final_output_conv_block.conv.weight",
              "dec_final_output_conv_block.bn.bias"]:
        out["w_" + k] = A(pd[k])
    out["buffers_untouched"] = np.array(all(np.array_equal(sd[k], A(v)) for k, v in net.named_buffers()))
    out["lr_last"] = np.array(mgr.optimizer.param_groups[0]["lr"])
    with open(os.path.join(HERE, "teacher_meta.json"), "w") as f:
        json.dump({"param_names": names}, f, indent=1)
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("teacher.npz", **out)


def gen_reorganize():
    """DynamicMBConvLayer.re_organize_middle_weights (dynamic_layers.py:156-199) and the net-level driver
    (ofa_mbs4.py:462-464) as supporting_elastic_expand uses them (progressive_shrinking.py:331-396): stage 0 on the
    det-filled weights, then stage 1 on the result.  Recorded: every tensor of the layer after each call, and the first
    and last MB block of an S4 net after net.re_organize_middle_weights(0) then (1)."""
    from ofa.elastic_nn.networks import OFAMobileNetS4
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    layer = DynamicMBConvLayer([16], [16], [3, 5, 7], [3, 4, 6], stride=1, act_func="relu6")
    _load_sd(layer, "reorg")
    out = {}
    for stage in (0, 1, 2):
        layer.re_organize_middle_weights(expand_ratio_stage=stage)
        for k, v in layer.state_dict().items():
            out["layer_s%d_%s" % (stage, k)] = A(v)
    net = OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                         pixelshuffle_depth_list=[1, 2])
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, "s4")
    net.load_state_dict({k: T(v) for k, v in sd.items()})
    for stage in (0, 1):
        net.re_organize_middle_weights(expand_ratio_stage=stage)
        for k, v in net.state_dict().items():
            if k.startswith("blocks.0.") or k.startswith("blocks.15."):
                out["net_s%d_%s" % (stage, k)] = A(v)
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    save("reorganize.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["pwconv", "dwconv", "bn", "pixelshuffle", "mbblock", "s4", "metric", "trainer", "calibration",
                             "div2k", "teacher", "reorganize"]
    for w in which:
        globals()["gen_" + w]()
