"""Pin the CPU oracle (oracle/) to fixtures generated from the reference itself
(tests/golden/make_golden.py).  The reference has no tests of its own,
so these goldens ARE the pin.  CPU only.

Tolerances: the reference ran fp32 oneDNN; the oracle accumulates in double.  Differences are
pure fp32 reassociation: rtol 2e-5 / atol 2e-6 on O(1) data.  PixelShuffle is bit-exact.
"""
import numpy as np
import pytest

from conftest import assert_close
from detfill import det_uniform, det_ints

RT, AT = 2e-5, 2e-6


def test_pixel_shuffle_bit_exact(ora, golden):
    g = golden("pixelshuffle.npz")
    for (N, C, H, W) in [(2, 3, 5, 7), (1, 16, 4, 4)]:
        tag = "%d_%d_%d_%d" % (N, C, H, W)
        x = det_ints((N, C * 4, H, W), "ps/x%d_%d" % (C, H), -64, 64)
        y = ora.pixel_shuffle(x, 2)
        assert y.tobytes() == g["shuffle_y_" + tag].tobytes()
        # round trip and unshuffle-vs-reference (the reference's one-hot conv on integers is exact)
        assert ora.pixel_unshuffle(y, 2).tobytes() == x.tobytes()
        z = det_ints((N, C, H * 2, W * 2), "pus/x%d_%d" % (C, H), -64, 64)
        assert ora.pixel_unshuffle(z, 2).tobytes() == g["unshuffle_y_" + tag].tobytes()
        # other element sizes move the same bytes
        for dt in (np.float16, np.int8, np.float64):
            xd = x.astype(dt)
            assert np.array_equal(ora.pixel_shuffle(xd, 2), g["shuffle_y_" + tag].astype(dt))


def test_pixel_shuffle_index_law(ora):
    # out[n,c,h*r+i,w*r+j] = in[n,c*r*r+i*r+j,h,w]  (SURVEY.md 8a row a8), r=2 and r=3
    for r in (2, 3):
        x = np.arange(2 * 2 * r * r * 3 * 5, dtype=np.float32).reshape(2, 2 * r * r, 3, 5)
        y = ora.pixel_shuffle(x, r)
        for n, c, h, w, i, j in [(0, 0, 0, 0, 0, 0), (1, 1, 2, 4, r - 1, r - 1), (0, 1, 1, 3, 1, 0)]:
            assert y[n, c, h * r + i, w * r + j] == x[n, c * r * r + i * r + j, h, w]


@pytest.mark.parametrize("oc", [192, 256, 384])
def test_pwconv_expand(ora, golden, oc):
    g = golden("pwconv.npz")
    x, w = g["expand_x"], g["expand_w"]
    y = ora.pwconv_fwd(x, w, oc)
    assert_close(y, g["expand_y_%d" % oc], RT, AT, "y")
    dy = det_uniform(y.shape, "pw/expand/dy%d" % oc)
    dx, dw = ora.pwconv_bwd(dy, x, w)
    assert_close(dx, g["expand_dx_%d" % oc], RT, AT, "dx")
    assert_close(dw, g["expand_dw_%d" % oc], RT, 2e-5, "dw")
    assert np.all(dw[oc:] == 0)  # exact zeros outside the slice


@pytest.mark.parametrize("ic", [192, 256, 384])
def test_pwconv_project_strided_slice(ora, golden, ic):
    g = golden("pwconv.npz")
    w = g["project_w"]
    x = det_uniform((2, ic, 6, 7), "pw/project/x%d" % ic)
    y = ora.pwconv_fwd(x, w, 64)
    assert_close(y, g["project_y_%d" % ic], RT, AT, "y")
    dy = det_uniform(y.shape, "pw/project/dy%d" % ic)
    dx, dw = ora.pwconv_bwd(dy, x, w)
    assert_close(dx, g["project_dx_%d" % ic], RT, AT, "dx")
    assert_close(dw, g["project_dw_%d" % ic], RT, 2e-5, "dw")
    assert np.all(dw[:, ic:] == 0)


@pytest.mark.parametrize("mode", [None, 1])
@pytest.mark.parametrize("C", [16, 24])
@pytest.mark.parametrize("k", [3, 5, 7])
def test_dwconv_and_kernel_transform(ora, golden, mode, C, k):
    g = golden("dwconv.npz")
    tag = "m%s_c%d_k%d" % ("N" if mode is None else "1", C, k)
    mats = None if mode is None else {"7to5": g["m75"], "5to3": g["m53"]}
    f = ora.ktransform_fwd(g["w7"], C, k, [3, 5, 7], mats)
    assert_close(f, g["filter_" + tag], RT, AT, "filter")
    x = g["x_%d" % C]
    y = ora.dwconv_fwd(x, f)
    assert_close(y, g["y_" + tag], RT, AT, "y")
    dy = det_uniform(y.shape, "dw/dy/" + tag)
    dx, df = ora.dwconv_bwd(dy, x, f)
    assert_close(dx, g["dx_" + tag], RT, AT, "dx")
    dw7, dm = ora.ktransform_bwd(df, g["w7"], C, k, [3, 5, 7], mats)
    assert_close(dw7, g["dw7_" + tag], RT, 2e-5, "dw7")
    # gradient sparsity: zero outside rows < C and outside the centre window
    assert np.all(dw7[C:] == 0)
    s0 = 3 - k // 2
    mask = np.ones((7, 7), bool)
    if mode is None or k == 7:
        mask[s0:s0 + k, s0:s0 + k] = False
    else:
        mask[1:6, 1:6] = False  # transform chain always enters through the 5x5 crop
    assert np.all(dw7[:, 0][:, mask] == 0)
    if mode is not None:
        # None-ness of the matrix gradients
        assert bool(g["dm75_isnone_" + tag]) == ("7to5" not in dm)
        assert bool(g["dm53_isnone_" + tag]) == ("5to3" not in dm)
        if "7to5" in dm:
            assert_close(dm["7to5"], g["dm75_" + tag], RT, 2e-5, "dm75")
        if "5to3" in dm:
            assert_close(dm["5to3"], g["dm53_" + tag], RT, 2e-5, "dm53")


def test_identity_matrices_give_centre_crops(ora, golden):
    # matrices are initialised to identity (dynamic_op.py:40) => transformed == cropped filters
    g = golden("dwconv.npz")
    eye = {"7to5": np.eye(25, dtype=np.float32), "5to3": np.eye(9, dtype=np.float32)}
    for k in (3, 5, 7):
        a = ora.ktransform_fwd(g["w7"], 24, k, [3, 5, 7], eye)
        b = ora.ktransform_fwd(g["w7"], 24, k, [3, 5, 7], None)
        assert np.array_equal(a, b)
        s0 = 3 - k // 2
        assert np.array_equal(b[:, 0], g["w7"][:24, 0, s0:s0 + k, s0:s0 + k])


@pytest.mark.parametrize("C", [16, 24])
@pytest.mark.parametrize("training", [True, False])
def test_sliced_batchnorm(ora, golden, C, training):
    from detfill import fill_state_dict
    g = golden("bn.npz")
    sd = fill_state_dict({"bn.weight": (24,), "bn.bias": (24,), "bn.running_mean": (24,),
                          "bn.running_var": (24,)}, "bnfix")
    rm, rv = sd["bn.running_mean"].copy(), sd["bn.running_var"].copy()
    x = det_uniform((3, C, 5, 6), "bn/x%d" % C, -2.0, 2.0)
    y, _, _ = ora.bn_fwd(x, sd["bn.weight"], sd["bn.bias"], rm, rv, training)
    tag = "c%d_%s" % (C, "train" if training else "eval")
    assert_close(y, g["y_" + tag], 2e-5, 5e-6, "y")
    assert_close(rm, g["rm_" + tag], 1e-6, 1e-7, "running_mean")
    assert_close(rv, g["rv_" + tag], 1e-6, 1e-7, "running_var")
    # [C:] untouched
    assert np.array_equal(rm[C:], sd["bn.running_mean"][C:])
    if training:
        dy = det_uniform(y.shape, "bn/dy%d" % C)
        dx, dg, db = ora.bn_bwd_train(dy, x, sd["bn.weight"])
        assert_close(dx, g["dx_" + tag], 5e-5, 5e-6, "dx")
        assert_close(dg, g["dgamma_" + tag][:C], 2e-5, 2e-5, "dgamma")
        assert_close(db, g["dbeta_" + tag][:C], 2e-5, 2e-5, "dbeta")
        assert np.all(g["dgamma_" + tag][C:] == 0)
        # num_batches_tracked: module path when dim==max increments too; manual path +1
        assert int(g["nbt_" + tag]) == 1


def test_metric_psnr_y(ora, golden):
    g = golden("metric.npz")
    assert np.array_equal(ora.tensor2img_u8(g["b"]), g["u8_b"])
    assert np.array_equal(ora.rgb2y(ora.tensor2img_u8(g["b"])), g["y_b"])
    assert abs(ora.psnr_y(g["a"], g["b"]) - float(g["psnr_ab"])) < 1e-9


def test_dense_conv_matches_torch(ora):
    # static ConvLayer conv (ofa/layers.py:131-151): oracle vs the same ATen op the reference calls
    import torch
    import torch.nn.functional as F
    x = det_uniform((2, 5, 7, 9), "cv/x")
    w = det_uniform((6, 5, 5, 5), "cv/w", -0.2, 0.2)
    dy = det_uniform((2, 6, 7, 9), "cv/dy")
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    y = F.conv2d(xt, wt, None, 1, 2)
    y.backward(torch.from_numpy(dy))
    assert_close(ora.conv2d_fwd(x, w), y.detach().numpy(), RT, AT, "y")
    dx, dw = ora.conv2d_bwd(dy, x, w)
    assert_close(dx, xt.grad.numpy(), RT, AT, "dx")
    assert_close(dw, wt.grad.numpy(), RT, 2e-5, "dw")
