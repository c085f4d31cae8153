"""GPU parity of the one-kernel eval-mode MB block (ofasr_mbconv_infer, csrc/mbfused.hip: BN folded into the
convolutions, the mid tensor never reaches HBM) against the CPU oracle's restatement of the same block
(oracle/composite16.py fused_eval_block: the C oracle's operators with double accumulation on the same folded 16-bit
operands), and against the reference's fp32 semantics (oracle/s4_port.py block, double) in the L2 norm.  `-m gpu`.

Reference: DynamicMBConvLayer.forward + shortcut in eval mode (dynamic_layers.py:70-84, dynamic_op.py:148-167 with
bn.training False, proxyless_nets.py:44-51).  Shapes: the BASELINE tile-aligned sizes, ragged Set14-like sizes with
odd widths (tiles cut by the image border, unaligned rows), single-tile images, every (mid, K)."""
import numpy as np
import pytest
import torch

from conftest import amd
from test_hip_composite16 import H, RT, _make_block, close16

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # (N, H, W, expand, K, dtype)
    (2, 64, 64, 6, 7, torch.bfloat16), (2, 64, 64, 6, 5, torch.bfloat16), (2, 64, 64, 6, 3, torch.bfloat16),
    (2, 64, 64, 4, 7, torch.bfloat16), (2, 64, 64, 3, 5, torch.bfloat16), (2, 64, 64, 3, 3, torch.bfloat16),
    (3, 32, 32, 4, 7, torch.bfloat16), (1, 48, 48, 6, 3, torch.bfloat16),
    (1, 30, 31, 6, 7, torch.bfloat16), (1, 45, 62, 4, 5, torch.bfloat16), (2, 17, 20, 3, 3, torch.bfloat16),
    (1, 16, 16, 6, 7, torch.bfloat16), (1, 5, 9, 6, 5, torch.bfloat16), (1, 33, 125, 6, 7, torch.bfloat16),
    (2, 64, 64, 6, 7, torch.float16), (2, 64, 64, 4, 5, torch.float16), (1, 30, 31, 3, 3, torch.float16),
    (1, 36, 44, 6, 5, torch.float16),
    # widths around the vector-request limits (W < 4: element-wise window; W < 8: element-wise rows), the sizes the tile
    # picker sends to 8x32 (125x90) and 4x64 (24x128), a 4x64 tile cut by the right border mid-quad (W = 70)
    (1, 9, 3, 3, 3, torch.bfloat16), (1, 9, 4, 3, 5, torch.bfloat16), (1, 6, 7, 4, 7, torch.bfloat16),
    (1, 12, 8, 3, 3, torch.bfloat16), (1, 125, 90, 3, 7, torch.bfloat16), (1, 24, 128, 4, 5, torch.bfloat16),
    (1, 13, 70, 3, 7, torch.float16),
    # more tiles than half the CUs: one workgroup per tile walks every chunk (the cases above have few tiles, so their
    # chunks are spread over several workgroups whose partial projections are folded by the last to arrive)
    (9, 64, 64, 3, 3, torch.bfloat16), (5, 64, 128, 3, 5, torch.float16),
]
# every tile shape on sizes it would not be picked for (ofasr_debug_mbfused_tile)
TILE_CASES = [(tw, c) for tw in (16, 32, 64) for c in (
    (2, 64, 64, 3, 7, torch.bfloat16), (1, 45, 62, 4, 5, torch.bfloat16), (1, 30, 31, 3, 3, torch.float16),
    (1, 21, 77, 3, 5, torch.bfloat16))]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_%dx%d_e%d_k%d_%s" % (
    c[0], c[1], c[2], c[3], c[4], "bf16" if c[5] == torch.bfloat16 else "f16"))
def test_fused_eval_block_vs_oracle(ora, case):
    _check_case(ora, case, None)


@pytest.mark.parametrize("tw,case", TILE_CASES, ids=lambda v: str(v) if isinstance(v, int) else "%dx%d_k%d" % (v[1], v[2], v[4]))
def test_fused_eval_block_every_tile_shape(ora, tw, case):
    lib = amd("_C").lib()
    prev = lib.ofasr_debug_mbfused_tile(tw)
    try:
        _check_case(ora, case, "%d, %d>" % (256 // tw, tw))
    finally:
        lib.ofasr_debug_mbfused_tile(prev)


def test_fused_eval_block_split_equals_unsplit_semantics(ora):
    """the same small call with and without spreading the chunks: both against the oracle, and close to each other
    (the summation order of the projection differs: per-split partial sums instead of one running accumulator)"""
    lib = amd("_C").lib()
    case = (2, 32, 64, 6, 7, torch.bfloat16)
    prev = lib.ofasr_debug_mbfused_split(0)
    try:
        y0 = _check_case(ora, case, None)
    finally:
        lib.ofasr_debug_mbfused_split(prev)
    y1 = _check_case(ora, case, None)
    d = (y0.double() - y1.double()).abs().max().item()
    assert d <= 2.0 ** -6 * max(1.0, y0.abs().max().item()), d    # a bf16 ulp or two of the output


def _check_case(ora, case, want_shape):
    from oracle import composite16 as c16
    from oracle import s4_port
    N, Hh, Ww, e, K, dtype = case
    C = amd("_C")
    block, layer = _make_block(300 + 10 * K + e)
    block.to(DEV).eval()
    layer.active_kernel_size, layer.active_expand_ratio = K, e
    mid = layer.active_middle_channel(64)
    sd0 = {k: v.detach().cpu().clone() for k, v in block.state_dict().items()}
    g = torch.Generator().manual_seed(19)
    x16 = torch.randn((N, 64, Hh, Ww), generator=g).to(dtype)
    C.reset_launch_counts()
    with torch.no_grad():
        y = block(x16.to(DEV))
    torch.cuda.synchronize()
    table = C.launch_table()
    assert sum(n for k, n in table.items() if k.startswith("mb_fused_kernel")) == 1, table
    if want_shape:
        assert any(k.startswith("mb_fused_kernel") and k.endswith(want_shape) for k in table), table
    assert not any(k.startswith("pw_") or k.startswith("dw_") or k.startswith("bn_") for k in table), table
    # nothing was written to the BN buffers
    for k, v in block.state_dict().items():
        assert torch.equal(v.cpu(), sd0[k]), k

    pfx = "mobile_inverted_conv."
    w1 = sd0[pfx + "inverted_bottleneck.conv.conv.weight"].numpy()
    w2 = sd0[pfx + "point_linear.conv.conv.weight"].numpy()
    wdw = sd0[pfx + "depth_conv.conv.conv.weight"].numpy()
    mats = {"7to5": sd0[pfx + "depth_conv.conv.7to5_matrix"].numpy(), "5to3": sd0[pfx + "depth_conv.conv.5to3_matrix"].numpy()}
    bn = {i: {k: sd0[pfx + nm + k].numpy() for k in ("weight", "bias", "running_mean", "running_var")}
          for i, nm in enumerate(("inverted_bottleneck.bn.bn.", "depth_conv.bn.bn.", "point_linear.bn.bn."))}
    f = ora.ktransform_fwd(wdw, mid, K, [3, 5, 7], mats)
    ref = c16.fused_eval_block(x16.float().numpy(), w1, f, w2, bn, mid, dtype)
    close16(H(y), ref, dtype, "fused eval block", frac_ok=2e-4)

    # and the reference semantics in double (no folding, no 16-bit intermediates): 16-bit realisation error
    sd = {"blocks.0." + k: v.double() for k, v in sd0.items()}
    yr = s4_port._mb_block(sd, "blocks.0.mobile_inverted_conv.", x16.double(), K, e, [3, 5, 7], True, False, 0.1, 1e-5)
    rel = float((y.double().cpu() - yr).norm() / yr.norm())
    assert rel <= (8e-3 if dtype == torch.bfloat16 else 1.2e-3), rel
    return y.float().cpu()


def test_fused_eval_block_is_used_by_the_network_in_eval_mode():
    """OFAMobileNetS4 in eval mode under autocast: every active MB block runs the fused kernel; the output equals the
    composite (un-fused) eval path's to 16-bit round-off; with gradients enabled the composite path is taken."""
    ops, C = amd("ops"), amd("_C")
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    torch.manual_seed(4)
    net = amd("elastic_nn.networks").OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6],
                                                    depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    net.init_model("he_fout")
    net.to(DEV).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.7, 1.3)
    net.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)
    x = torch.rand(2, 3, 40, 48, device=DEV)
    C.reset_launch_counts()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    n_active = sum(1 for kind, _ in net.active_block_sequence() if kind == "mb")
    assert C.launch_count("mb_fused_kernel") == n_active and n_active >= 7
    was = ops.FUSED_INFER
    ops.FUSED_INFER = False
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y0 = net(x)
    finally:
        ops.FUSED_INFER = was
    rel = float((y.float() - y0.float()).norm() / y0.float().norm())
    assert rel <= 1.5e-2, rel
    C.reset_launch_counts()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net(x)
    assert C.launch_count("mb_fused_kernel") == 0


def test_fused_eval_block_operands_prepared_once():
    """the BN-folded operand images are a function of the weights only: a second eval-mode forward of the same block
    launches neither the kernel transform nor the fold kernel and returns the same bits; an optimizer-style in-place
    update, a re-organisation of the middle channels and a return to training mode each invalidate them."""
    C, ops = amd("_C"), amd("ops")
    block, layer = _make_block(77)
    block.to(DEV).eval()
    layer.active_kernel_size, layer.active_expand_ratio = 5, 4
    x = torch.randn(2, 64, 24, 40, device=DEV).bfloat16()
    ops.clear_infer_cache()

    def run():
        C.reset_launch_counts()
        with torch.no_grad():
            y = block(x)
        return y, C.launch_count("mb_fold_kernel"), C.launch_count("kt_fwd_kernel"), C.launch_count("mb_fused_kernel")

    y0, nf, nk, nm = run()
    assert (nf, nk, nm) == (1, 1, 1)
    y1, nf, nk, nm = run()
    assert (nf, nk, nm) == (0, 0, 1) and torch.equal(y0, y1)
    with torch.no_grad():
        layer.depth_conv.conv.conv.weight.add_(0.05)
    y2, nf, nk, nm = run()
    assert (nf, nk, nm) == (1, 1, 1) and not torch.equal(y0, y2)
    layer.re_organize_middle_weights()
    _, nf, _, _ = run()
    assert nf == 1
    block.train()
    block.eval()
    _, nf2, _, _ = run()
    y3, nf3, _, _ = run()
    assert nf2 == 1 and nf3 == 0
    # another active kernel size of the same block is another set of operands
    layer.active_kernel_size = 3
    _, nf, nk, _ = run()
    assert (nf, nk) == (1, 1)
