"""Stage-by-stage CPU oracle of the composite MB block with 16-bit activations.

TEST INFRASTRUCTURE ONLY (see the header of ofasr_oracle.c): imported by tests/ as the checker of
`ofasr_mbconv_fwd/_bwd` (include/ofasr.h), never by the product path.

The block is the reference's DynamicMBConvLayer.forward + identity shortcut
(/root/reference ofa/elastic_nn/modules/dynamic_layers.py:70-84, ofa/elastic_nn/modules/dynamic_op.py:46-84,
104-112,148-167, ofa/imagenet_codebase/networks/proxyless_nets.py:44-51).  With 16-bit activations the HIP path
stores y1 (expand), y2 (depthwise), y3 (project), out and, in the backward, dy3, dy2, dy1, dx as 16-bit tensors, and
applies BN1/BN2 + ReLU6 while READING y1 / y2 (the activated tensors are never stored).  Each function below restates
ONE stage with the C oracle's operators (double accumulation, oracle/ofasr_oracle.c) fed the 16-bit tensors the stage
reads, so a test can hand it the tensors the GPU stage actually read and compare what the GPU stage wrote: the
only differences left are fp32-vs-double accumulation and the final rounding.

`mma=True` marks operands that go through the matrix cores and are therefore rounded to the activation type first
(1x1 weights always; the depthwise input and filter only on the Toeplitz/MFMA depthwise kernel, k in {5,7}).
"""
import numpy as np
import torch

from . import oracle

RELU6_LO, RELU6_HI = 0.0, 6.0


def r16(a, dtype):
    """round an fp32 / fp64 array to the 16-bit activation type (RNE, as v_cvt_pk_bf16_f32 / v_cvt_f16_f32) -> fp32"""
    if dtype == torch.float32:
        return np.asarray(a, np.float32)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype).float().numpy()


def bn_consts(y, gamma, beta, rm, rv, training, eps=1e-5):
    """per-channel (mean, invstd) in double: batch statistics of `y` (training) or the running statistics"""
    C = y.shape[1]
    if training:
        yd = y.astype(np.float64)
        mean = yd.mean(axis=(0, 2, 3))
        var = yd.var(axis=(0, 2, 3))
    else:
        mean, var = rm[:C].astype(np.float64), rv[:C].astype(np.float64)
    return mean, 1.0 / np.sqrt(var + eps)


def bn_apply(y, mean, invstd, gamma, beta, act):
    C = y.shape[1]
    g = gamma[:C].astype(np.float64).reshape(1, C, 1, 1)
    b = beta[:C].astype(np.float64).reshape(1, C, 1, 1)
    pre = (y.astype(np.float64) - mean.reshape(1, C, 1, 1)) * (g * invstd.reshape(1, C, 1, 1)) + b
    return (np.clip(pre, RELU6_LO, RELU6_HI) if act else pre), pre


def running_update(y, rm, rv, momentum=0.1):
    """nn.BatchNorm2d's running-statistics update on the first C channels (dynamic_op.py:157-167)"""
    C = y.shape[1]
    yd = y.astype(np.float64)
    n = yd.size / C
    rm2, rv2 = rm.astype(np.float64).copy(), rv.astype(np.float64).copy()
    rm2[:C] = (1 - momentum) * rm2[:C] + momentum * yd.mean(axis=(0, 2, 3))
    rv2[:C] = (1 - momentum) * rv2[:C] + momentum * yd.var(axis=(0, 2, 3)) * n / max(n - 1, 1)
    return rm2, rv2


def bn_bwd(dy, y, mean, invstd, gamma, pre, act, training):
    """gradient of act(BN(y)) w.r.t. y, gamma, beta (double); `pre` = BN output before the activation"""
    C = y.shape[1]
    dz = dy.astype(np.float64)
    if act:
        dz = dz * ((pre > RELU6_LO) & (pre < RELU6_HI))
    xhat = (y.astype(np.float64) - mean.reshape(1, C, 1, 1)) * invstd.reshape(1, C, 1, 1)
    db = dz.sum(axis=(0, 2, 3))
    dg = (dz * xhat).sum(axis=(0, 2, 3))
    k = (gamma[:C].astype(np.float64) * invstd).reshape(1, C, 1, 1)
    if training:
        M = dz.size / C
        dx = k * (dz - (db / M).reshape(1, C, 1, 1) - xhat * (dg / M).reshape(1, C, 1, 1))
    else:
        dx = k * dz
    return dx, dg, db


def edge_safe(pre, margin):
    """elements whose ReLU6 mask cannot flip under a perturbation of `margin` of the pre-activation"""
    return (np.abs(pre - RELU6_LO) > margin) & (np.abs(pre - RELU6_HI) > margin)


def expand_fwd(x16, w1_full, mid, dtype):
    return oracle.pwconv_fwd(x16, r16(w1_full, dtype), mid)


def depthwise_fwd(a1, f, dtype, mma):
    return oracle.dwconv_fwd(r16(a1, dtype) if mma else a1.astype(np.float32), r16(f, dtype) if mma else f)


def project_fwd(a2, w2_full, cout, dtype):
    return oracle.pwconv_fwd(r16(a2, dtype), r16(w2_full, dtype), cout)


# ------------------------------------------------------------------- the eval-mode block with BN folded (one kernel)
def folded_operands(w1_full, f, w2_full, bn, mid, dtype, eps=1e-5):
    """the operands ofasr_mbconv_infer computes with: eval-mode BN i is y -> y*s_i + t_i with
    s = gamma / sqrt(running_var + eps), t = beta - running_mean * s (dynamic_op.py:148-167 in eval mode), folded into
    the rows of the 1x1 weights and into the depthwise taps, 16-bit operands / fp32 biases.
    bn: {0,1,2: {weight, bias, running_mean, running_var}}; f: active filter [mid,1,K,K]."""
    def st(i, C):
        s = bn[i]["weight"][:C].astype(np.float64) / np.sqrt(bn[i]["running_var"][:C].astype(np.float64) + eps)
        return s.astype(np.float32), (bn[i]["bias"][:C].astype(np.float64) - bn[i]["running_mean"][:C] * s).astype(np.float32)

    s1, t1 = st(0, mid)
    s2, t2 = st(1, mid)
    cout = w2_full.shape[0]
    s3, t3 = st(2, cout)
    cin = w1_full.shape[1]
    w1f = r16(w1_full[:mid, :cin, 0, 0] * s1[:, None], dtype)
    ff = r16(np.asarray(f, np.float32).reshape(mid, -1) * s2[:, None], dtype).reshape(np.asarray(f).shape)
    w2f = r16(w2_full[:cout, :mid, 0, 0] * s3[:, None], dtype)
    return w1f, t1, ff, t2, w2f, t3


def fused_eval_block(x16, w1_full, f, w2_full, bn, mid, dtype, residual=True):
    """restatement of the fused eval-mode block with the C oracle's operators (double accumulation): 16-bit a1 / a2
    (they are matrix-core / dot-product operands), fp32 everywhere else, one rounding of the output."""
    w1f, t1, ff, t2, w2f, t3 = folded_operands(w1_full, f, w2_full, bn, mid, dtype)
    y1 = oracle.pwconv_fwd(x16, w1f.reshape(mid, -1, 1, 1), mid) + t1.reshape(1, mid, 1, 1)
    a1 = r16(np.clip(y1, RELU6_LO, RELU6_HI), dtype)
    y2 = oracle.dwconv_fwd(a1, ff) + t2.reshape(1, mid, 1, 1)
    a2 = r16(np.clip(y2, RELU6_LO, RELU6_HI), dtype)
    cout = w2f.shape[0]
    y3 = oracle.pwconv_fwd(a2, w2f.reshape(cout, mid, 1, 1), cout) + t3.reshape(1, cout, 1, 1)
    return r16(y3.astype(np.float64) + (x16 if residual else 0.0), dtype)
