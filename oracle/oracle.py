"""numpy front-end of the CPU oracle (oracle/ofasr_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of ofasr_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, as the checker.

All arrays are C-contiguous float32 NCHW unless noted; shapes follow the reference's tensors
(ofa/elastic_nn/modules/dynamic_op.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ofasr_oracle.c")
_LIB = os.path.join(_HERE, "libofasr_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds).  Building the checker is not using it."""
    if (not force) and os.path.exists(_LIB) and (
        (not os.path.exists(_SRC)) or os.path.getmtime(_LIB) >= os.path.getmtime(_SRC)
    ):
        return _LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-o", _LIB, _SRC, "-lm"]
    subprocess.check_call(cmd)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_L = ctypes.c_long
_I = ctypes.c_int
_D = ctypes.c_double


# ----------------------------------------------------------------------------- pixel shuffle
def pixel_shuffle(x, r=2):
    """x [N, C*r*r, H, W] -> [N, C, H*r, W*r]; any dtype, bit-exact (ofa/utils.py:309-310)."""
    x = np.ascontiguousarray(x)
    N, Cr, H, W = x.shape
    C = Cr // (r * r)
    assert C * r * r == Cr
    y = np.empty((N, C, H * r, W * r), dtype=x.dtype)
    lib().ora_pixel_shuffle_fwd(_p(x), _p(y), _L(N), _L(C), _L(H), _L(W), _I(r), _I(x.itemsize))
    return y


def pixel_unshuffle(x, r=2):
    """x [N, C, H*r, W*r] -> [N, C*r*r, H, W]; any dtype, bit-exact (ofa/utils.py:383-397)."""
    x = np.ascontiguousarray(x)
    N, C, Ho, Wo = x.shape
    H, W = Ho // r, Wo // r
    assert H * r == Ho and W * r == Wo
    y = np.empty((N, C * r * r, H, W), dtype=x.dtype)
    lib().ora_pixel_unshuffle_fwd(_p(x), _p(y), _L(N), _L(C), _L(H), _L(W), _I(r), _I(x.itemsize))
    return y


# --------------------------------------------------------------------------------- pointwise
def pwconv_fwd(x, w_full, cout):
    """x [N,Cin,H,W], w_full [Cout_max, Cin_max, 1, 1] -> y [N,cout,H,W] (dynamic_op.py:104-112)."""
    x = _f32(x)
    w = _f32(w_full)
    N, Cin, H, W = x.shape
    ldw = w.shape[1]
    y = np.empty((N, cout, H, W), np.float32)
    lib().ora_pwconv_fwd(_p(x), _p(w), _L(ldw), _p(y), _L(N), _L(Cin), _L(cout), _L(H * W))
    return y


def pwconv_bwd(dy, x, w_full):
    """returns (dx, dw_full) -- dw_full has the parameter's full shape, zeros outside the slice."""
    dy = _f32(dy)
    x = _f32(x)
    w = _f32(w_full)
    N, Cin, H, W = x.shape
    cout = dy.shape[1]
    ldw = w.shape[1]
    dx = np.empty_like(x)
    dw = np.zeros_like(w)
    lib().ora_pwconv_dgrad(_p(dy), _p(w), _L(ldw), _p(dx), _L(N), _L(Cin), _L(cout), _L(H * W))
    lib().ora_pwconv_wgrad(_p(dy), _p(x), _p(dw), _L(ldw), _L(N), _L(Cin), _L(cout), _L(H * W))
    return dx, dw


# --------------------------------------------------------------------------------- depthwise
def dwconv_fwd(x, f):
    """x [N,C,H,W], f [C,1,K,K] or [C,K,K] -> y (dynamic_op.py:79-83)."""
    x = _f32(x)
    f = _f32(f)
    N, C, H, W = x.shape
    K = f.shape[-1]
    y = np.empty_like(x)
    lib().ora_dwconv_fwd(_p(x), _p(f), _p(y), _L(N), _L(C), _L(H), _L(W), _I(K))
    return y


def dwconv_bwd(dy, x, f):
    """returns (dx, df) with df shaped like f."""
    dy = _f32(dy)
    x = _f32(x)
    f = _f32(f)
    N, C, H, W = x.shape
    K = f.shape[-1]
    dx = np.empty_like(x)
    df = np.empty_like(f)
    lib().ora_dwconv_dgrad(_p(dy), _p(f), _p(dx), _L(N), _L(C), _L(H), _L(W), _I(K))
    lib().ora_dwconv_wgrad(_p(dy), _p(x), _p(df), _L(N), _L(C), _L(H), _L(W), _I(K))
    return dx, df


# ---------------------------------------------------------------------------- kernel transform
def _chain(ks_set, K):
    """kernel sizes walked from max(ks_set) down to K, as the loop at dynamic_op.py:54-69 does."""
    ks_sorted = sorted(set(ks_set))
    chain = [s for s in reversed(ks_sorted) if s >= K]
    assert chain[-1] == K, "active kernel %d not in %s" % (K, ks_sorted)
    return chain


def _mat_ptrs(arrs):
    arr_t = ctypes.c_void_p * max(1, len(arrs))
    return arr_t(*[a.ctypes.data for a in arrs]) if arrs else arr_t(None)


def ktransform_fwd(w_max, C, K, ks_set, mats=None):
    """w_max [Cmax,1,kmax,kmax]; mats: dict {'7to5': [25,25], '5to3': [9,9]} or None
    (KERNEL_TRANSFORM_MODE None) -> f [C,1,K,K]  (dynamic_op.py:46-71)."""
    w = _f32(w_max)
    kmax = w.shape[-1]
    chain = _chain(ks_set, K)
    assert chain[0] == kmax
    nsteps = len(chain) - 1
    ks = (ctypes.c_int * len(chain))(*chain)
    transform = 0 if mats is None else 1
    marr = []
    if transform:
        for s in range(nsteps):
            marr.append(_f32(mats["%dto%d" % (chain[s], chain[s + 1])]))
    f = np.empty((C, 1, K, K), np.float32)
    lib().ora_ktransform_fwd(_p(w), ks, _I(nsteps), _mat_ptrs(marr), _I(transform), _p(f), _L(C))
    return f


def ktransform_bwd(df, w_max, C, K, ks_set, mats=None):
    """returns (dw_max full-shape, {'7to5': dM, ...} for the steps actually walked)."""
    w = _f32(w_max)
    df = _f32(df)
    kmax = w.shape[-1]
    chain = _chain(ks_set, K)
    nsteps = len(chain) - 1
    ks = (ctypes.c_int * len(chain))(*chain)
    transform = 0 if mats is None else 1
    marr, darr, names = [], [], []
    if transform:
        for s in range(nsteps):
            name = "%dto%d" % (chain[s], chain[s + 1])
            names.append(name)
            marr.append(_f32(mats[name]))
            darr.append(np.zeros_like(marr[-1]))
    dw = np.zeros_like(w)
    lib().ora_ktransform_bwd(_p(w), ks, _I(nsteps), _mat_ptrs(marr), _I(transform), _p(df),
                             _p(dw), _mat_ptrs(darr), _L(C))
    return dw, dict(zip(names, darr))


# -------------------------------------------------------------------------------- dense conv
def conv2d_fwd(x, w):
    x = _f32(x)
    w = _f32(w)
    N, Cin, H, W = x.shape
    Cout, _, K, _ = w.shape
    y = np.empty((N, Cout, H, W), np.float32)
    lib().ora_conv2d_fwd(_p(x), _p(w), _p(y), _L(N), _L(Cin), _L(Cout), _L(H), _L(W), _I(K))
    return y


def conv2d_bwd(dy, x, w):
    dy = _f32(dy)
    x = _f32(x)
    w = _f32(w)
    N, Cin, H, W = x.shape
    Cout, _, K, _ = w.shape
    dx = np.empty_like(x)
    dw = np.empty_like(w)
    lib().ora_conv2d_dgrad(_p(dy), _p(w), _p(dx), _L(N), _L(Cin), _L(Cout), _L(H), _L(W), _I(K))
    lib().ora_conv2d_wgrad(_p(dy), _p(x), _p(dw), _L(N), _L(Cin), _L(Cout), _L(H), _L(W), _I(K))
    return dx, dw


# ---------------------------------------------------------------------------------- batchnorm
def bn_fwd(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5):
    """BatchNorm2d over the first C=x.shape[1] channels of the (possibly longer) parameter /
    buffer arrays (dynamic_op.py:148-167).  running_* are updated IN PLACE in training mode.
    returns (y, save_mean, save_invstd)."""
    x = _f32(x)
    N, C, H, W = x.shape
    y = np.empty_like(x)
    g = _f32(gamma[:C])
    b = _f32(beta[:C])
    rm = _f32(running_mean[:C]).copy()
    rv = _f32(running_var[:C]).copy()
    sm = np.empty(C, np.float32)
    si = np.empty(C, np.float32)
    lib().ora_bn_fwd(_p(x), _p(y), _p(g), _p(b), _p(rm), _p(rv), _I(1 if training else 0),
                     _D(momentum), _D(eps), _p(sm), _p(si), _L(N), _L(C), _L(H * W))
    if training:
        running_mean[:C] = rm
        running_var[:C] = rv
    return y, sm, si


def bn_bwd_train(dy, x, gamma, eps=1e-5):
    """returns (dx, dgamma[:C], dbeta[:C])."""
    dy = _f32(dy)
    x = _f32(x)
    N, C, H, W = x.shape
    g = _f32(gamma[:C])
    dx = np.empty_like(x)
    dg = np.empty(C, np.float32)
    db = np.empty(C, np.float32)
    lib().ora_bn_bwd_train(_p(dy), _p(x), _p(g), _D(eps), _p(dx), _p(dg), _p(db),
                           _L(N), _L(C), _L(H * W))
    return dx, dg, db


# ------------------------------------------------------------------------------------- metric
def tensor2img_u8(t):
    """batch-1 restatement of tensor2img_np (sr_run_manager.py:567-590): clamp to [0,1],
    CHW->HWC, *255, round-half-even (numpy .round()), uint8.  t: [1,3,H,W] or [3,H,W]."""
    a = np.asarray(t, dtype=np.float32)
    if a.ndim == 4:
        assert a.shape[0] == 1, "PSNR parity is defined for batch 1 (SURVEY.md Q10)"
        a = a[0]
    a = np.clip(a, 0.0, 1.0)
    a = np.transpose(a, (1, 2, 0))
    return (a * np.float32(255.0)).round().astype(np.uint8)


def rgb2y(img):
    """sr_run_manager.py:592-597 -- BT.601 luma on uint8 HWC, rounded, cast back to uint8."""
    assert img.dtype == np.uint8
    y = (np.dot(img[..., :3], [65.481, 128.553, 24.966]) / 255.0 + 16.0).round()
    return y.astype(np.uint8)


def psnr_u8(a, b):
    """ofa/utils.py:27-34."""
    assert a.dtype == b.dtype == np.uint8
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20.0 * np.log10(255.0 / np.sqrt(mse))


def psnr_y(out, ref):
    """the parity metric: psnr(rgb2y(tensor2img_np(out)), rgb2y(tensor2img_np(ref)))."""
    return psnr_u8(rgb2y(tensor2img_u8(out)), rgb2y(tensor2img_u8(ref)))
