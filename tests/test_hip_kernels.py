"""GPU parity tests proper: every HIP kernel, called through the C ABI (ops.py -> _C.py ->
libofasr_hip.so), against the CPU oracle on the same seeded inputs and against the goldens the
reference produced.  Run with `-m gpu` on an MI355X.

Tolerances
  fp32 activations : rtol 2e-5, atol 2e-6 * scale   (fp32 reassociation only; oracle sums in double)
  16-bit activations: inputs are rounded to the 16-bit type FIRST and that rounded tensor is
                     what the oracle sees; the kernel accumulates in fp32 and rounds once on
                     store => |err| <= 2^-8 (bf16) / 2^-11 (f16) relative, tested as rtol 1e-2 / 2e-3.
  PixelShuffle     : bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import amd, assert_close
from detfill import det_uniform, det_ints

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return amd("ops")


def G(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def H(t):
    return t.detach().float().cpu().numpy()


def tol(dtype, scale=1.0):
    if dtype == torch.float32:
        return dict(rtol=2e-5, atol=2e-6 * scale)
    if dtype == torch.bfloat16:
        return dict(rtol=1e-2, atol=1e-2 * scale)
    return dict(rtol=2e-3, atol=2e-3 * scale)


def rounded(a, dtype):
    """the value the GPU actually sees for a host fp32 array under `dtype`"""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


DTYPES = [torch.float32, torch.bfloat16, torch.float16]


# ------------------------------------------------------------------------------- pixel shuffle
@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (1, 16, 4, 4), (2, 64, 64, 64), (1, 5, 33, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16, torch.int8, torch.float64])
def test_pixel_shuffle_bit_exact(ops, ora, shape, dtype):
    N, C, Hh, W = shape
    x = det_ints((N, C * 4, Hh, W), "gps/%d_%d" % (C, Hh), -64, 64)
    xt = torch.from_numpy(x).to(dtype).to(DEV)
    y = ops.pixel_shuffle(xt, 2)
    ref = ora.pixel_shuffle(xt.cpu().view(torch.uint8).numpy().view(_np(dtype)), 2)
    assert y.cpu().view(torch.uint8).numpy().tobytes() == ref.tobytes()
    back = ops.pixel_unshuffle(y, 2)
    assert torch.equal(back, xt)


def _np(dtype):
    return {torch.float32: np.uint32, torch.bfloat16: np.uint16, torch.float16: np.uint16, torch.int8: np.uint8,
            torch.float64: np.uint64}[dtype]


def test_pixel_shuffle_golden_and_r3(ops, golden, ora):
    g = golden("pixelshuffle.npz")
    for (N, C, Hh, W) in [(2, 3, 5, 7), (1, 16, 4, 4)]:
        tag = "%d_%d_%d_%d" % (N, C, Hh, W)
        x = det_ints((N, C * 4, Hh, W), "ps/x%d_%d" % (C, Hh), -64, 64)
        assert H(ops.pixel_shuffle(G(x), 2)).tobytes() == g["shuffle_y_" + tag].tobytes()
        z = det_ints((N, C, Hh * 2, W * 2), "pus/x%d_%d" % (C, Hh), -64, 64)
        assert H(ops.pixel_unshuffle(G(z), 2)).tobytes() == g["unshuffle_y_" + tag].tobytes()
    x = det_ints((2, 2 * 9, 4, 5), "gps/r3", -64, 64)
    assert H(ops.pixel_shuffle(G(x), 3)).tobytes() == ora.pixel_shuffle(x, 3).tobytes()


def test_pixel_shuffle_backward_is_unshuffle(ops):
    x = G(det_uniform((2, 8, 6, 6), "gps/bw")).requires_grad_(True)
    y = ops.pixel_shuffle(x, 2)
    dy = G(det_uniform(tuple(y.shape), "gps/bw/dy"))
    y.backward(dy)
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert torch.equal(x.grad, torch.nn.functional.pixel_unshuffle(dy, 2))


def test_pixel_shuffle_full_size_roundtrip(ops):
    # BASELINE shapes: [16,256,64,64] -> [16,64,128,128] and [16,256,128,128] -> [16,64,256,256]
    for (N, C, Hh, W) in [(16, 64, 64, 64), (16, 64, 128, 128)]:
        x = torch.randint(-2 ** 31, 2 ** 31 - 1, (N, C * 4, Hh, W), dtype=torch.int32, device=DEV).view(torch.float32)
        y = ops.pixel_shuffle(x, 2)
        # size-independent properties: inverse round trip + index law on random probes
        assert torch.equal(ops.pixel_unshuffle(y, 2).view(torch.int32), x.view(torch.int32))
        idx = torch.randint(0, 2 ** 30, (64, 6))
        for n, c, h, w, i, j in idx.tolist():
            n, c, h, w, i, j = n % N, c % C, h % Hh, w % W, i % 2, j % 2
            assert y.view(torch.int32)[n, c, 2 * h + i, 2 * w + j] == x.view(torch.int32)[n, 4 * c + 2 * i + j, h, w]


# --------------------------------------------------------------------------- kernel transform
@pytest.mark.parametrize("mode", [None, 1])
@pytest.mark.parametrize("C", [16, 24])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_ktransform_golden(ops, golden, mode, C, k):
    g = golden("dwconv.npz")
    tag = "m%s_c%d_k%d" % ("N" if mode is None else "1", C, k)
    w7 = G(g["w7"]).requires_grad_(True)
    m75 = G(g["m75"]).requires_grad_(True)
    m53 = G(g["m53"]).requires_grad_(True)
    chain = [s for s in (7, 5, 3) if s >= k]
    mats = [] if (mode is None or k == 7) else ([m75] if k == 5 else [m75, m53])
    f = ops.KTransformFn.apply(w7, C, tuple(chain), mode is not None, *mats)
    assert_close(H(f), g["filter_" + tag], 2e-5, 2e-6, "filter")
    # feed the reference's filter gradient: recompute it from the golden dw conv
    x = g["x_%d" % C]
    dy = det_uniform(x.shape, "dw/dy/" + tag)
    from oracle import oracle
    _, df = oracle.dwconv_bwd(dy, x, g["filter_" + tag])
    f.backward(G(df))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert_close(H(w7.grad), g["dw7_" + tag], 2e-5, 2e-5, "dw7")
    if mats:
        assert_close(H(m75.grad), g["dm75_" + tag], 2e-5, 2e-5, "dm75")
        if k == 3:
            assert_close(H(m53.grad), g["dm53_" + tag], 2e-5, 2e-5, "dm53")
        else:
            assert m53.grad is None
    else:
        assert m75.grad is None and m53.grad is None


def test_ktransform_full_width(ops, ora):
    w7 = det_uniform((384, 1, 7, 7), "gkt/w7", -0.3, 0.3)
    m75 = (np.eye(25, dtype=np.float32) + det_uniform((25, 25), "gkt/m75", -0.2, 0.2))
    m53 = (np.eye(9, dtype=np.float32) + det_uniform((9, 9), "gkt/m53", -0.2, 0.2))
    for C in (192, 256, 384):
        for k in (3, 5):
            mats = {"7to5": m75, "5to3": m53}
            f_ref = ora.ktransform_fwd(w7, C, k, [3, 5, 7], mats)
            df = det_uniform(f_ref.shape, "gkt/df%d_%d" % (C, k))
            dw_ref, dm_ref = ora.ktransform_bwd(df, w7, C, k, [3, 5, 7], mats)
            wt = G(w7).requires_grad_(True)
            a, b = G(m75).requires_grad_(True), G(m53).requires_grad_(True)
            f = ops.KTransformFn.apply(wt, C, (7, 5) if k == 5 else (7, 5, 3), True, *([a] if k == 5 else [a, b]))
            assert_close(H(f), f_ref, 2e-5, 2e-6, "f")
            f.backward(G(df))
            amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
            assert_close(H(wt.grad), dw_ref, 2e-5, 2e-5, "dw")
            assert_close(H(a.grad), dm_ref["7to5"], 5e-5, 1e-4, "dm75")
            if k == 3:
                assert_close(H(b.grad), dm_ref["5to3"], 5e-5, 1e-4, "dm53")


# -------------------------------------------------------------------------------- depthwise
DW_SHAPES = [
    (2, 16, 9, 11),     # golden-sized, ragged
    (2, 24, 64, 64),    # north-star LR plane
    (1, 8, 48, 48),     # config[1] LR plane
    (1, 6, 40, 150),    # W > 64 -> strips with halo; H > 32 -> row chunks
    (3, 5, 1, 1),       # degenerate
    (1, 4, 7, 64),
    (1, 3, 70, 65),
    (2, 6, 12, 8),      # 2 lanes per row, 32 row groups per wave (X4 encoder resolution)
    (1, 4, 6, 4),       # 1 lane per row
    (2, 3, 5, 16),
    (1, 2, 40, 24),
]


@pytest.mark.parametrize("shape", DW_SHAPES)
@pytest.mark.parametrize("k", [1, 3, 5, 7])
@pytest.mark.parametrize("dtype", DTYPES)
def test_dwconv_vs_oracle(ops, ora, shape, k, dtype):
    N, C, Hh, W = shape
    x = rounded(det_uniform(shape, "gdw/x%s" % (shape,)), dtype)
    f = det_uniform((C, 1, k, k), "gdw/f%d_%d" % (C, k), -0.4, 0.4)
    dy = rounded(det_uniform(shape, "gdw/dy%s" % (shape,)), dtype)
    xt = G(x, dtype).requires_grad_(True)
    ft = G(f).requires_grad_(True)
    y = ops.dwconv(xt, ft)
    assert y.dtype == dtype
    y_ref = ora.dwconv_fwd(x, f)
    assert_close(H(y), y_ref, what="y", **tol(dtype, k))
    y.backward(G(dy, dtype))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    dx_ref, df_ref = ora.dwconv_bwd(dy, x, f)
    assert_close(H(xt.grad), dx_ref, what="dx", **tol(dtype, k))
    # df sums N*H*W products of exactly-representable inputs in fp32: fp32 tolerance for all dtypes
    scale = max(1.0, float(np.sqrt(N * Hh * W)))
    assert_close(H(ft.grad), df_ref, 1e-4, 2e-6 * scale, "df")


@pytest.mark.parametrize("mode", [None, 1])
@pytest.mark.parametrize("C", [16, 24])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_dwconv_golden(ops, golden, mode, C, k):
    g = golden("dwconv.npz")
    tag = "m%s_c%d_k%d" % ("N" if mode is None else "1", C, k)
    x = G(g["x_%d" % C]).requires_grad_(True)
    y = ops.dwconv(x, G(g["filter_" + tag]))
    assert_close(H(y), g["y_" + tag], 2e-5, 2e-6 * k, "y")
    y.backward(G(det_uniform(tuple(y.shape), "dw/dy/" + tag)))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert_close(H(x.grad), g["dx_" + tag], 2e-5, 2e-6 * k, "dx")


def test_dwconv_full_size_linearity(ops):
    # BASELINE size [16,384,64,64], k=7: conv is linear in x and in f (size-independent properties)
    torch.manual_seed(0)
    x1 = torch.randn(16, 384, 64, 64, device=DEV)
    x2 = torch.randn(16, 384, 64, 64, device=DEV)
    f = torch.randn(384, 1, 7, 7, device=DEV) * 0.1
    a = ops.dwconv(x1, f) + ops.dwconv(x2, f)
    b = ops.dwconv(x1 + x2, f)
    assert float((a - b).abs().max()) < 2e-4
    # a delta filter at the centre is the identity
    d = torch.zeros(384, 1, 7, 7, device=DEV)
    d[:, 0, 3, 3] = 1.0
    assert torch.equal(ops.dwconv(x1, d), x1)
    # shifting taps: delta at (i,j) == shifted image with zero fill
    d.zero_()
    d[:, 0, 0, 6] = 1.0
    y = ops.dwconv(x1, d)  # y[h,w] = x[h-3, w+3]
    assert torch.equal(y[:, :, 3:, :61], x1[:, :, :61, 3:])
    assert float(y[:, :, :3].abs().max()) == 0 and float(y[:, :, :, 61:].abs().max()) == 0


def test_dwconv_full_size_16bit_matrix_core_path(ops):
    """BASELINE size [16,384,64,64], k=7, bf16: the depthwise conv runs as Toeplitz GEMMs on the matrix cores; delta
    filters make it an exact identity / shift (pins the swizzled plane image, the band fragments and the transposed
    write-back at full size), forward and input gradient."""
    torch.manual_seed(1)
    x = torch.randn(16, 384, 64, 64, device=DEV).to(torch.bfloat16)
    d = torch.zeros(384, 1, 7, 7, device=DEV)
    d[:, 0, 3, 3] = 1.0
    assert torch.equal(ops.dwconv(x, d), x)
    d.zero_()
    d[:, 0, 1, 5] = 1.0          # y[h, w] = x[h - 2, w + 2]
    xr = x.clone().requires_grad_(True)
    y = ops.dwconv(xr, d)
    assert torch.equal(y[:, :, 2:, :62], x[:, :, :62, 2:])
    assert float(y[:, :, :2].abs().max()) == 0 and float(y[:, :, :, 62:].abs().max()) == 0
    g = torch.randn_like(y)
    y.backward(g)                # dx[h, w] = g[h + 2, w - 2]
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert torch.equal(xr.grad[:, :, :62, 2:], g[:, :, 2:, :62])
    assert float(xr.grad[:, :, 62:].abs().max()) == 0 and float(xr.grad[:, :, :, :2].abs().max()) == 0


# -------------------------------------------------------------------------------- pointwise
PW_CASES = [
    # (N, Cin, Cout_active, Cin_max, Cout_max, H, W)
    (2, 64, 192, 64, 384, 6, 7),      # expand, ragged HW=42 (unaligned path)
    (2, 64, 384, 64, 384, 16, 16),    # expand aligned
    (2, 64, 256, 64, 384, 12, 10),    # HW=120: aligned for 16-bit (8 | 120), tail tile
    (2, 192, 64, 384, 64, 6, 7),      # project from a strided row slice, ragged
    (2, 384, 64, 384, 64, 16, 16),    # project aligned
    (1, 256, 64, 384, 64, 12, 10),
    (1, 3, 5, 8, 8, 5, 5),            # tiny / odd everything
    (1, 100, 70, 128, 96, 9, 8),      # K > 64 and M > 64: generic fan-in over 2 row passes
    (1, 64, 64, 64, 64, 64, 64),      # one full LR image
    (2, 64, 256, 64, 384, 16, 16),    # expand to 2 whole 128-row slabs on whole 128-pixel tiles: slab-walk kernel, NSLAB = 2
    (2, 256, 64, 384, 64, 16, 16),    # its project (input gradient = the same kernel with K = 64)
    (1, 64, 192, 64, 384, 16, 16),    # 1.5 slabs: generic fan-out kernel with branch-free requests, full and partial waves
]


@pytest.mark.parametrize("case", PW_CASES)
@pytest.mark.parametrize("dtype", DTYPES)
def test_pwconv_vs_oracle(ops, ora, case, dtype):
    N, Cin, Cout, Cin_max, Cout_max, Hh, W = case
    x = rounded(det_uniform((N, Cin, Hh, W), "gpw/x%s" % (case,)), dtype)
    a = float(np.sqrt(3.0 / Cin))
    w = det_uniform((Cout_max, Cin_max, 1, 1), "gpw/w%s" % (case,), -a, a)
    dy = rounded(det_uniform((N, Cout, Hh, W), "gpw/dy%s" % (case,)), dtype)
    xt = G(x, dtype).requires_grad_(True)
    wt = G(w).requires_grad_(True)
    y = ops.pwconv(xt, wt, Cout)
    assert y.dtype == dtype and tuple(y.shape) == (N, Cout, Hh, W)
    # 16-bit paths round the fp32 master weights to the activation type inside the kernel
    w_eff = w if dtype == torch.float32 else rounded(w, dtype)
    y_ref = ora.pwconv_fwd(x, w_eff, Cout)
    assert_close(H(y), y_ref, what="y", **tol(dtype))
    y.backward(G(dy, dtype))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    dx_ref, _ = ora.pwconv_bwd(dy, x, w_eff)
    _, dw_ref = ora.pwconv_bwd(dy, x, w)
    assert_close(H(xt.grad), dx_ref, what="dx", **tol(dtype, np.sqrt(Cout / Cin) if Cout > Cin else 1.0))
    scale = max(1.0, float(np.sqrt(N * Hh * W)))
    assert_close(H(wt.grad), dw_ref, 1e-4, 2e-6 * scale, "dw")
    g = H(wt.grad)
    assert np.all(g[Cout:] == 0) and np.all(g[:, Cin:] == 0)


@pytest.mark.parametrize("oc", [192, 256, 384])
def test_pwconv_golden_expand(ops, golden, oc):
    g = golden("pwconv.npz")
    x = G(g["expand_x"]).requires_grad_(True)
    w = G(g["expand_w"]).requires_grad_(True)
    y = ops.pwconv(x, w, oc)
    assert_close(H(y), g["expand_y_%d" % oc], 2e-5, 2e-6, "y")
    y.backward(G(det_uniform(tuple(y.shape), "pw/expand/dy%d" % oc)))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert_close(H(x.grad), g["expand_dx_%d" % oc], 2e-5, 2e-6, "dx")
    assert_close(H(w.grad), g["expand_dw_%d" % oc], 2e-5, 2e-5, "dw")


@pytest.mark.parametrize("ic", [192, 256, 384])
def test_pwconv_golden_project(ops, golden, ic):
    g = golden("pwconv.npz")
    x = G(det_uniform((2, ic, 6, 7), "pw/project/x%d" % ic)).requires_grad_(True)
    w = G(g["project_w"]).requires_grad_(True)
    y = ops.pwconv(x, w, 64)
    assert_close(H(y), g["project_y_%d" % ic], 2e-5, 2e-6, "y")
    y.backward(G(det_uniform(tuple(y.shape), "pw/project/dy%d" % ic)))
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    assert_close(H(x.grad), g["project_dx_%d" % ic], 2e-5, 2e-6, "dx")
    assert_close(H(w.grad), g["project_dw_%d" % ic], 2e-5, 2e-5, "dw")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pwconv_full_size_properties(ops, dtype):
    # BASELINE size: N=16, 64 <-> 384 channels at 64x64.  Identity-like weights make the GEMM a copy /
    # channel gather, which pins the pixel<->MFMA-column permutation at every position exactly.
    torch.manual_seed(1)
    x = torch.randn(16, 64, 64, 64, device=DEV).to(dtype)
    w = torch.zeros(384, 64, 1, 1, device=DEV)
    perm = torch.randperm(64, device=DEV)
    for r in range(384):
        w[r, perm[r % 64]] = 1.0
    y = ops.pwconv(x, w, 384)
    assert torch.equal(y, x[:, perm[torch.arange(384, device=DEV) % 64]])
    # project: selecting channels out of 384
    x2 = torch.randn(16, 384, 64, 64, device=DEV).to(dtype)
    w2 = torch.zeros(64, 384, 1, 1, device=DEV)
    sel = torch.randperm(384, device=DEV)[:64]
    w2[torch.arange(64, device=DEV), sel] = 1.0
    assert torch.equal(ops.pwconv(x2, w2, 64), x2[:, sel])
    # linearity in x on random weights
    wr = torch.randn(384, 64, 1, 1, device=DEV) * 0.1
    xa, xb = x, torch.randn_like(x.float()).to(dtype)
    lhs = ops.pwconv(xa, wr, 384).float() + ops.pwconv(xb, wr, 384).float()
    rhs = ops.pwconv((xa.float() + xb.float()).to(dtype), wr, 384).float()
    assert float((lhs - rhs).abs().max()) < (1e-4 if dtype == torch.float32 else 0.25)
