"""GPU parity of the implicit-GEMM dense convolution (C ABI ofasr_conv2d_fwd/_dgrad) against the CPU oracle's
direct convolution (oracle/ofasr_oracle.c ora_conv2d_*, itself checked against the ATen op the reference calls)."""
import numpy as np
import pytest
import torch

from conftest import amd, assert_close
from detfill import det_uniform

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _n_conv(C, f32=False):
    """forward / input-gradient launches of the static-conv kernels: the implicit-GEMM kernel of the dtype, or -- for a
    3-channel side (stem / head) -- the thin-side kernels of csrc/conv_thin.hip"""
    return C.launch_count("conv_f32_kernel" if f32 else "conv_igemm_kernel") + C.launch_count("ct_out_kernel") + \
        C.launch_count("ct_in_kernel")


def _n_wgrad(C, f32=False):
    return C.launch_count("conv_f32_wgrad_kernel" if f32 else "conv_wgrad_kernel") + C.launch_count("ct_wg_kernel")

CASES = [
    # N, Cin, Cout, H, W, K
    (1, 64, 128, 6, 64, 5),     # one 128-row slab, 3 row tiles
    (2, 64, 256, 5, 16, 5),     # two slabs, odd H, W < tile width
    (1, 3, 64, 4, 72, 5),       # stem: 3 input channels (one k-step), ragged second tile
    (1, 64, 3, 7, 64, 5),       # head: 3 output channels (32-row slab)
    (1, 64, 64, 4, 128, 5),     # 64-row slab
    (1, 128, 64, 4, 64, 3),     # 3x3, two channel chunks
    (2, 16, 40, 9, 24, 3),
    (3, 64, 160, 6, 64, 3),     # wgrad: two 128-row slabs
    (2, 96, 24, 4, 16, 5),      # wgrad: two input-channel slabs, thin output
    (2, 64, 3, 6, 24, 3),       # 3x3 head (X4 nets): csrc/conv_thin.hip with k = 3
    (2, 3, 64, 5, 40, 3),       # 3x3 stem
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv2d_vs_oracle(ora, case, dtype):
    ops = amd("ops")
    N, Cin, Cout, H, W, K = case
    r16 = lambda a: torch.from_numpy(a).to(dtype).float().numpy()
    x = r16(det_uniform((N, Cin, H, W), "cv2/x%s" % (case,)))
    a = float(np.sqrt(3.0 / (Cin * K * K)))
    w = det_uniform((Cout, Cin, K, K), "cv2/w%s" % (case,), -a, a)
    dy = r16(det_uniform((N, Cout, H, W), "cv2/dy%s" % (case,)))
    xt = torch.from_numpy(x).to(dtype).to(DEV).requires_grad_(True)
    wt = torch.from_numpy(w).to(DEV).requires_grad_(True)
    y = ops.Conv2dFn.apply(xt, wt)
    y_ref = ora.conv2d_fwd(x, r16(w))
    rt = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert_close(y.detach().float().cpu().numpy(), y_ref, rt, rt, "y")
    y.backward(torch.from_numpy(dy).to(dtype).to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    dx_ref, dw_ref = ora.conv2d_bwd(dy, x, r16(w))
    scale = float(np.sqrt(Cout * K * K / max(Cin * K * K, 1)))
    assert_close(xt.grad.float().cpu().numpy(), dx_ref, rt, rt * max(1.0, scale), "dx")
    # weight gradient: fp32 accumulation of exact 16-bit products, fixed summation order
    assert_close(wt.grad.cpu().numpy(), dw_ref, 1e-3, 1e-3 * float(np.abs(dw_ref).max()), "dw")


def test_conv_layer_uses_hip_conv_under_autocast():
    ops = amd("ops")
    layers = amd("layers")
    torch.manual_seed(0)
    layer = layers.ConvLayer(64, 256, kernel_size=5, act_func="pixelshuffle", use_bn=True).to(DEV).train()
    x = torch.randn(2, 64, 8, 64, device=DEV)
    outs = []
    for hip in (True, False):
        ops.HIP_CONV = hip
        ops.CONV_FORCE_HIP = hip
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(x)
        finally:
            ops.HIP_CONV = True
            ops.CONV_FORCE_HIP = False
        outs.append(y.float())
    assert outs[0].shape == (2, 64, 16, 128)
    assert float((outs[0] - outs[1]).detach().abs().max()) < 0.08


@pytest.mark.parametrize("shape", [(16, 64, 256, 128, 128, 5), (16, 64, 3, 256, 256, 5), (16, 3, 64, 64, 64, 5)])
def test_conv2d_full_size_properties(shape):
    """BASELINE-size layers (the two up-sampler convs' big sibling, the head, the stem): size-independent checks.
    (1) a one-hot kernel (centre tap, output channel m reads input channel m % Cin) makes the convolution a channel
    selection -- exact in 16 bits, pins the pixel <-> MFMA-column permutation and the window addressing at full size;
    (2) adjoint identities <dy, conv(x; w)> = <dgrad(dy; w), x> = <wgrad(dy, x), w> tie the three kernels together."""
    ops = amd("ops")
    N, Cin, Cout, H, W, K = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    x = (torch.randint(-8, 9, (N, Cin, H, W), generator=g).float() / 8).to(torch.bfloat16).to(DEV)
    w = torch.zeros(Cout, Cin, K, K, device=DEV)
    for m in range(Cout):
        w[m, m % Cin, K // 2, K // 2] = 1.0
    y = ops.Conv2dFn.apply(x, w)
    assert torch.equal(y, x[:, [m % Cin for m in range(Cout)]])
    # adjoint identities with small-integer data (every product and partial sum exact in fp32 up to the final sums)
    w2 = (torch.randint(-2, 3, (Cout, Cin, K, K), generator=g).float() / 4).to(DEV).requires_grad_(True)
    xs = x[:2].clone().requires_grad_(True)
    dy = (torch.randint(-4, 5, (2, Cout, H, W), generator=g).float() / 4).to(torch.bfloat16).to(DEV)
    y2 = ops.Conv2dFn.apply(xs, w2)
    y2.backward(dy)
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    lhs = float((dy.double() * y2.detach().double()).sum())
    mid = float((xs.grad.double() * xs.detach().double()).sum())
    rhs = float((w2.grad.double() * w2.detach().double()).sum())
    scale = float((dy.double().abs() * y2.detach().double().abs()).sum()) + 1e-9
    # y2 and dx are rounded to bf16 on store (relative 2^-9 per element, random sign): the sums agree far tighter
    assert abs(lhs - mid) <= 2e-3 * scale and abs(lhs - rhs) <= 2e-3 * scale and abs(mid - rhs) <= 2e-3 * scale


@pytest.mark.parametrize("case", [(1, 64, 64, 9, 125, 5), (2, 3, 64, 7, 62, 5), (1, 64, 256, 5, 31, 5), (1, 64, 3, 6, 146, 5),
                                  (1, 16, 16, 5, 13, 3)])
def test_conv2d_ragged_width_vs_oracle(ora, case):
    """widths that are not a multiple of 8 (Set14 LR sizes: 125, 62, 146 ...): ops.conv2d zero-pads on the right to
    the kernel's 16-byte row granularity and drops the extra columns -- the HIP kernel serves the layer (no vendor
    fall-back), output and gradients against the oracle conv on the unpadded tensors."""
    ops, C = amd("ops"), amd("_C")
    N, Cin, Cout, H, W, K = case
    dtype = torch.bfloat16
    r16 = lambda a: torch.from_numpy(a).to(dtype).float().numpy()
    x = r16(det_uniform((N, Cin, H, W), "cvr/x%s" % (case,)))
    a = float(np.sqrt(3.0 / (Cin * K * K)))
    w = det_uniform((Cout, Cin, K, K), "cvr/w%s" % (case,), -a, a)
    dy = r16(det_uniform((N, Cout, H, W), "cvr/dy%s" % (case,)))
    conv = torch.nn.Conv2d(Cin, Cout, K, padding=K // 2, bias=False).to(DEV)
    conv.weight.data.copy_(torch.from_numpy(w))
    xt = torch.from_numpy(x).to(dtype).to(DEV).requires_grad_(True)
    C.reset_launch_counts()
    y = ops.conv2d(xt, conv)
    assert _n_conv(C) == 1 and tuple(y.shape) == (N, Cout, H, W)
    y_ref = ora.conv2d_fwd(x, r16(w))
    assert_close(y.detach().float().cpu().numpy(), y_ref, 1e-2, 1e-2, "y")
    y.backward(torch.from_numpy(dy).to(dtype).to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    dx_ref, dw_ref = ora.conv2d_bwd(dy, x, r16(w))
    scale = float(np.sqrt(Cout * K * K / max(Cin * K * K, 1)))
    assert_close(xt.grad.float().cpu().numpy(), dx_ref, 1e-2, 1e-2 * max(1.0, scale), "dx")
    assert_close(conv.weight.grad.cpu().numpy(), dw_ref, 1e-3, 1e-3 * float(np.abs(dw_ref).max()), "dw")


@pytest.mark.parametrize("case", CASES + [(1, 64, 3, 6, 4, 3),     # X4 encoder tail at 6 x 4: narrower than a wide fragment of the thin weight gradient
                                          (1, 64, 64, 9, 125, 5), (2, 3, 64, 7, 62, 5), (1, 16, 16, 5, 13, 3),
                                          (2, 64, 256, 16, 16, 5), (1, 40, 70, 6, 33, 3),
                                          # weight gradient with loader waves (K = 5, W % 4 == 0): ragged channels, a 4-column
                                          # second tile and a 2-row last tile; then 1 or 2 tiles per block (both LDS buffers)
                                          (1, 40, 70, 6, 36, 5), (3, 64, 64, 64, 64, 5)])
def test_conv2d_fp32_vs_oracle(ora, case):
    """fp32 activations (the reference's arithmetic) on the fp32 matrix instruction (csrc/conv2d_f32.hip): forward,
    input and weight gradients against the oracle conv (double accumulation) at fp32 tolerances, any width; the layer
    is served by the library (no vendor kernel)."""
    ops, C = amd("ops"), amd("_C")
    N, Cin, Cout, H, W, K = case
    x = det_uniform((N, Cin, H, W), "cvf/x%s" % (case,))
    a = float(np.sqrt(3.0 / (Cin * K * K)))
    w = det_uniform((Cout, Cin, K, K), "cvf/w%s" % (case,), -a, a)
    dy = det_uniform((N, Cout, H, W), "cvf/dy%s" % (case,))
    conv = torch.nn.Conv2d(Cin, Cout, K, padding=K // 2, bias=False).to(DEV)
    conv.weight.data.copy_(torch.from_numpy(w))
    xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
    C.reset_launch_counts()
    was, ops.CONV_FORCE_HIP = ops.CONV_FORCE_HIP, True   # the own kernels whatever the measured policy says for the shape
    try:
        y = ops.conv2d(xt, conv)
    finally:
        ops.CONV_FORCE_HIP = was
    assert _n_conv(C, True) == 1 and y.dtype == torch.float32
    y_ref = ora.conv2d_fwd(x, w)
    assert_close(y.detach().cpu().numpy(), y_ref, 5e-5, 5e-6, "y")
    y.backward(torch.from_numpy(dy).to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert _n_conv(C, True) == 2 and _n_wgrad(C, True) == 1
    dx_ref, dw_ref = ora.conv2d_bwd(dy, x, w)
    scale = float(np.sqrt(Cout * K * K / max(Cin * K * K, 1)))
    assert_close(xt.grad.cpu().numpy(), dx_ref, 5e-5, 5e-6 * max(1.0, scale), "dx")
    assert_close(conv.weight.grad.cpu().numpy(), dw_ref, 1e-4, 2e-6 * float(np.abs(dw_ref).max()) * np.sqrt(N * H * W), "dw")


INFER_CASES = [
    # N, Cin, Cout, H, W, K, act_func, use_bn
    (2, 64, 256, 6, 64, 5, "pixelshuffle", True),    # decoder stage: conv -> BN -> PixelShuffle(2) in one kernel
    (1, 64, 256, 5, 125, 5, "pixelshuffle", True),   # ragged Set14 width
    (1, 3, 64, 7, 72, 5, "relu6", True),             # stem
    (2, 64, 64, 4, 128, 5, None, True),
    (1, 64, 3, 6, 146, 5, None, False),              # head without BN, ragged
    (1, 64, 128, 5, 40, 3, "relu", True),            # an activation the epilogue does not know: applied afterwards
]


@pytest.mark.parametrize("case", INFER_CASES, ids=lambda c: "%dx%d_%dto%d_k%d_%s%s" % (c[3], c[4], c[1], c[2], c[5], c[6],
                                                                                       "" if c[7] else "_nobn"))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_layer_inference_one_kernel_vs_oracle(ora, case, dtype):
    """eval-mode ConvLayer = conv -> BN(running statistics) -> ReLU6 | PixelShuffle(2) (reference ofa/layers.py:120-151,
    ofa_mbs4.py:111-123) as ONE kernel (ofasr_conv2d_infer_run) against the oracle's conv / BN / PixelShuffle in
    double on the same 16-bit inputs and 16-bit-rounded weights; one rounding of the result."""
    layers, C = amd("layers"), amd("_C")
    N, Cin, Cout, H, W, K, act, use_bn = case
    r16 = lambda a: torch.from_numpy(a).to(dtype).float().numpy()
    x = r16(det_uniform((N, Cin, H, W), "cvi/x%s" % (case,)))
    a = float(np.sqrt(3.0 / (Cin * K * K)))
    w = det_uniform((Cout, Cin, K, K), "cvi/w%s" % (case,), -a, a)
    layer = layers.ConvLayer(Cin, Cout, kernel_size=K, use_bn=use_bn, act_func=act).to(DEV).eval()
    layer.conv.weight.data.copy_(torch.from_numpy(w))
    if use_bn:
        g = det_uniform((Cout,), "cvi/g%s" % (case,), 0.5, 1.5)
        b = det_uniform((Cout,), "cvi/b%s" % (case,), -0.5, 0.5)
        rm = det_uniform((Cout,), "cvi/rm%s" % (case,), -0.3, 0.3)
        rv = det_uniform((Cout,), "cvi/rv%s" % (case,), 0.2, 1.2)
        for t, v in ((layer.bn.weight, g), (layer.bn.bias, b), (layer.bn.running_mean, rm), (layer.bn.running_var, rv)):
            t.data.copy_(torch.from_numpy(v))
    amd("ops").clear_infer_cache()
    xt = torch.from_numpy(x).to(dtype).to(DEV)
    C.reset_launch_counts()
    with torch.no_grad():
        y = layer(xt)
    table = C.launch_table()
    assert C.launch_count("conv_igemm_kernel") == 1 and C.launch_count("conv_prep_kernel") == 1, table
    assert C.launch_count("bn_") == 0 and C.launch_count("ps_") == 0, table

    ref = ora.conv2d_fwd(x, r16(w)).astype(np.float64)
    if use_bn:
        sc = g.astype(np.float64) / np.sqrt(rv.astype(np.float64) + layer.bn.eps)
        ref = ref * sc.reshape(1, -1, 1, 1) + (b - rm * sc).reshape(1, -1, 1, 1)
    if act == "relu6":
        ref = np.clip(ref, 0.0, 6.0)
    elif act == "relu":
        ref = np.maximum(ref, 0.0)
    elif act == "pixelshuffle":
        ref = ora.pixel_shuffle(ref.astype(np.float32), 2).astype(np.float64)
    assert tuple(y.shape) == ref.shape
    rt = 1e-2 if dtype == torch.bfloat16 else 2e-3
    assert_close(y.float().cpu().numpy(), ref.astype(np.float32), rt, rt, "y")

    # second call: the prepared operands are reused (no weight-image launch), same bits
    C.reset_launch_counts()
    with torch.no_grad():
        y2 = layer(xt)
    assert C.launch_count("conv_prep_kernel") == 0 and C.launch_count("conv_igemm_kernel") == 1
    assert torch.equal(y, y2)
    # a tracked in-place write to a weight invalidates them
    with torch.no_grad():
        layer.conv.weight.mul_(2.0)
    C.reset_launch_counts()
    with torch.no_grad():
        y3 = layer(xt)
    assert C.launch_count("conv_prep_kernel") == 1
    if not use_bn and act is None:
        assert_close(y3.float().cpu().numpy(), 2.0 * ref.astype(np.float32), rt, 2 * rt, "y after the weight update")


def test_conv2d_fp32_policy_routes_and_matches_oracle(ora):
    """_conv_f32_policy: a regular training shape takes the own forward / input gradient and the vendor weight gradient;
    a 3-channel conv the thin-side kernels (csrc/conv_thin.hip) for all three; a ragged width the own kernels for all
    three -- same results either way."""
    ops, C = amd("ops"), amd("_C")
    was, ops.CONV_F32_VENDOR = ops.CONV_F32_VENDOR, True     # the opt-in mix (OFASR_CONV_F32_VENDOR=1); default: own kernels
    try:
        _policy_cases(ora, ops, C)
    finally:
        ops.CONV_F32_VENDOR = was


def _policy_cases(ora, ops, C):
    for case, want in (((2, 64, 128, 8, 64, 5), (1, 1, 0)), ((1, 64, 3, 8, 64, 5), (0, 0, 0)), ((1, 64, 128, 8, 60, 5), (1, 1, 1))):
        N, Cin, Cout, H, W, K = case
        x = det_uniform((N, Cin, H, W), "cvp/x%s" % (case,))
        a = float(np.sqrt(3.0 / (Cin * K * K)))
        w = det_uniform((Cout, Cin, K, K), "cvp/w%s" % (case,), -a, a)
        dy = det_uniform((N, Cout, H, W), "cvp/dy%s" % (case,))
        conv = torch.nn.Conv2d(Cin, Cout, K, padding=K // 2, bias=False).to(DEV)
        conv.weight.data.copy_(torch.from_numpy(w))
        xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
        C.reset_launch_counts()
        y = ops.conv2d(xt, conv)
        y.backward(torch.from_numpy(dy).to(DEV))
        amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
        assert C.launch_count("conv_f32_kernel") == want[0] + want[1], (case, C.launch_table())
        assert C.launch_count("conv_f32_wgrad_kernel") == want[2], (case, C.launch_table())
        if min(Cin, Cout) <= 4:
            assert C.launch_count("ct_out_kernel") == 1 and C.launch_count("ct_in_kernel") == 1 and \
                C.launch_count("ct_wg_kernel") == 1, (case, C.launch_table())
        dx_ref, dw_ref = ora.conv2d_bwd(dy, x, w)
        assert_close(y.detach().cpu().numpy(), ora.conv2d_fwd(x, w), 5e-5, 5e-6, "y")
        scale = float(np.sqrt(Cout * K * K / max(Cin * K * K, 1)))
        assert_close(xt.grad.cpu().numpy(), dx_ref, 5e-5, 5e-6 * max(1.0, scale), "dx")
        assert_close(conv.weight.grad.cpu().numpy(), dw_ref, 1e-4, 2e-6 * float(np.abs(dw_ref).max()) * np.sqrt(N * H * W), "dw")


TRAIN_EPI_CASES = [
    # N, Cin, Cout, H, W, K, act_func
    (2, 64, 256, 6, 64, 5, "pixelshuffle"),   # decoder stage: statistics in the conv epilogue, PixelShuffle in the BN apply's store
    (3, 3, 64, 8, 72, 5, "relu6"),            # stem: tiles cut by the right border (72 = 64 + 8)
    (2, 64, 64, 5, 128, 5, None),
    (2, 64, 128, 9, 32, 3, "relu6"),          # odd H: the last row of the last 2-row strip is outside the image
    (2, 64, 3, 12, 136, 5, None),             # head: 3 output channels (csrc/conv_thin.hip), two column strips
]


@pytest.mark.parametrize("case", TRAIN_EPI_CASES, ids=lambda c: "%dx%d_%dto%d_k%d_%s" % (c[3], c[4], c[1], c[2], c[5], c[6]))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv_layer_training_epilogue_statistics_vs_oracle(ora, case, dtype):
    """training-mode ConvLayer (reference ofa/layers.py:120-151): the conv kernel's epilogue takes the BatchNorm statistics
    (ofasr_conv2d_fwd_stat -> ofasr_bn_fwd_cp / ofasr_bn_finalize_cp + ofasr_pixel_shuffle2_bn).  Forward, running
    statistics, and every gradient (dX, dW, dgamma, dbeta through ofasr_bn_bwd_ps2 / ofasr_bn_act_bwd and the conv
    backward kernels) against the oracle's conv -> BN(batch statistics of the 16-bit conv output) -> act chain in
    double; no statistics pass and no shuffle kernel in the forward, no un-shuffle pass in the backward."""
    layers, C, ops = amd("layers"), amd("_C"), amd("ops")
    N, Cin, Cout, H, W, K, act = case
    r16 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype).float().numpy()
    x = r16(det_uniform((N, Cin, H, W), "cvt/x%s" % (case,)))
    a = float(np.sqrt(3.0 / (Cin * K * K)))
    w = det_uniform((Cout, Cin, K, K), "cvt/w%s" % (case,), -a, a)
    g = det_uniform((Cout,), "cvt/g%s" % (case,), 0.5, 1.5)
    b = det_uniform((Cout,), "cvt/b%s" % (case,), -0.5, 0.5)
    layer = layers.ConvLayer(Cin, Cout, kernel_size=K, use_bn=True, act_func=act).to(DEV).train()
    with torch.no_grad():
        layer.conv.weight.copy_(torch.from_numpy(w))
        layer.bn.weight.copy_(torch.from_numpy(g))
        layer.bn.bias.copy_(torch.from_numpy(b))
    xt = torch.from_numpy(x).to(dtype).to(DEV).requires_grad_(True)
    C.reset_launch_counts()
    y = layer(xt)
    table = C.launch_table()
    assert _n_conv(C) == 1 and C.launch_count("bn_stats_kernel") == 0, table
    assert C.launch_count("ps_r2_kernel") == 0 and C.launch_count("ps_generic") == 0, table
    assert C.launch_count("ps_r2_bn_kernel") == (1 if act == "pixelshuffle" else 0), table

    yc = r16(ora.conv2d_fwd(x, r16(w)))                       # the conv output as stored
    rm, rv = np.zeros(Cout, np.float32), np.ones(Cout, np.float32)
    z, _, _ = ora.bn_fwd(yc, g, b, rm, rv, True)
    ref = z.astype(np.float64)
    if act == "relu6":
        ref = np.clip(ref, 0.0, 6.0)
    elif act == "pixelshuffle":
        ref = ora.pixel_shuffle(ref.astype(np.float32), 2).astype(np.float64)
    rt = 1e-2 if dtype == torch.bfloat16 else 2e-3
    # the GPU's conv output may differ from the oracle's by one 16-bit rounding, which BN scales by gamma / sigma
    sig = np.sqrt(yc.astype(np.float64).var(axis=(0, 2, 3)).min())
    assert_close(y.detach().float().cpu().numpy(), ref.astype(np.float32), rt, rt * max(1.0, 1.5 / sig), "y")
    assert_close(layer.bn.running_mean.cpu().numpy(), rm, 1e-3, 1e-4, "running_mean")
    assert_close(layer.bn.running_var.cpu().numpy(), rv, 2e-3, 1e-4, "running_var")
    assert int(layer.bn.num_batches_tracked) == 1

    dy = r16(det_uniform(tuple(y.shape), "cvt/dy%s" % (case,)))
    C.reset_launch_counts()
    y.backward(torch.from_numpy(dy).to(dtype).to(DEV))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    if act == "pixelshuffle":   # the BN backward reads the gradient through the inverse shuffle: no un-shuffle pass
        assert C.launch_count("ps_r2_kernel") == 0 and C.launch_count("ps_generic") == 0, C.launch_table()
        assert C.launch_count("bn_bwd_reduce_ps_kernel") == 1 and C.launch_count("bn_bwd_apply_ps_kernel") == 1
    got = [xt.grad.float().cpu().numpy(), layer.conv.weight.grad.cpu().numpy(), layer.bn.weight.grad.cpu().numpy(),
           layer.bn.bias.grad.cpu().numpy()]

    # (1) backward against the ORACLE, stage by stage on the 16-bit tensors each GPU stage read (reference
    # ofa/layers.py:120-151 backward = PixelShuffle^-1 -> ReLU6 mask -> BatchNorm backward (batch statistics) -> conv
    # backward).  The BN backward (ofasr_bn_bwd_ps2 / ofasr_bn_act_bwd) read the incoming gradient and the conv output
    # the GPU stored; the conv backward read the 16-bit dyc that BN backward stored.
    with torch.no_grad():
        yc_gpu = ops.Conv2dFn.apply(torch.from_numpy(x).to(dtype).to(DEV), layer.conv.weight.detach()).float().cpu().numpy()
    assert_close(yc_gpu, yc, rt, rt, "stored conv output")
    dz = dy.astype(np.float32)
    if act == "pixelshuffle":
        dz = ora.pixel_unshuffle(dz, 2)
    elif act == "relu6":
        z_gpu, _, _ = ora.bn_fwd(yc_gpu, g, b, np.zeros(Cout, np.float32), np.ones(Cout, np.float32), True)
        dz = dz * ((z_gpu > 0.0) & (z_gpu < 6.0))
        # an element whose pre-activation sits within rounding of 0 or 6 may fall on the other side of the mask
        edge = (np.abs(z_gpu) < 1e-3) | (np.abs(z_gpu - 6.0) < 1e-3)
        assert edge.mean() < 1e-2
    dyc_ref, dg_ref, db_ref = ora.bn_bwd_train(dz, yc_gpu, g)
    big = lambda a_: max(float(np.abs(a_).max()), 1e-3)
    assert_close(got[2], dg_ref, 2e-3, 2e-3 * big(dg_ref), "dgamma vs oracle")
    assert_close(got[3], db_ref, 2e-3, 2e-3 * big(db_ref), "dbeta vs oracle")
    dx_ref, dw_ref = ora.conv2d_bwd(r16(dyc_ref), x, r16(w))
    # dx / dw: the GPU's stored dyc is the oracle's up to one 16-bit rounding per element (relative rt / 4), which the
    # convolution sums over Cout * K * K (dx) and N * H * W (dw) terms with random signs
    assert_close(got[0], dx_ref, rt, rt * big(dx_ref), "dx vs oracle")
    assert_close(got[1], dw_ref, rt, rt * big(dw_ref), "dw vs oracle")

    # (2) and against the un-fused HIP path on the same layer (conv -> statistics pass -> apply -> shuffle kernel)
    for p in (layer.conv.weight, layer.bn.weight, layer.bn.bias):
        p.grad = None
    with torch.no_grad():
        layer.bn.running_mean.zero_()
        layer.bn.running_var.fill_(1.0)
    was, ops.CONV_BN_EPILOGUE = ops.CONV_BN_EPILOGUE, False
    try:
        x2 = torch.from_numpy(x).to(dtype).to(DEV).requires_grad_(True)
        y2 = layer(x2)
        y2.backward(torch.from_numpy(dy).to(dtype).to(DEV))
        amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    finally:
        ops.CONV_BN_EPILOGUE = was
    assert_close(y.detach().float().cpu().numpy(), y2.detach().float().cpu().numpy(), rt, rt, "y vs the un-fused path")
    ref_g = [x2.grad.float().cpu().numpy(), layer.conv.weight.grad.cpu().numpy(), layer.bn.weight.grad.cpu().numpy(),
             layer.bn.bias.grad.cpu().numpy()]
    for name, a_, b_ in zip(("dx", "dw", "dgamma", "dbeta"), got, ref_g):
        assert_close(a_, b_, 2 * rt, 2 * rt * max(float(np.abs(b_).max()), 1e-3), name)
