"""torch.autograd bindings of the HIP hot-path kernels (C ABI: include/ofasr.h via _C.py).

Each Function replaces one ATen call site of the reference (file:line in the docstrings) with a
hand-written gfx950 kernel for forward AND backward.  PyTorch is plumbing here: it owns device
memory, the stream and the autograd graph; all arithmetic of these ops happens in
csrc/*.hip.  There is no CPU / eager fallback: tensors must live on the GPU.
"""
import collections
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C

_DT = {torch.float32: _C.F32, torch.float16: _C.F16, torch.bfloat16: _C.BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _C.OfasrError("unsupported activation dtype %s (f32/f16/bf16 only)" % t.dtype)


def _gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _C.OfasrError(
                "OFA-SR hot-path ops run only on the MI355X HIP kernels: got a %s tensor (no CPU fallback)" % t.device)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current stream of the current device as a raw handle (every library call takes one: this runs ~600 times per
    training step, and building a torch.cuda.Stream object each time cost several microseconds of host time per call)"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _ws(nbytes, device):
    n = max(int(nbytes), 16)
    t = torch.empty(n, dtype=torch.uint8, device=device)
    return t, ctypes.c_void_p(t.data_ptr()), ctypes.c_size_t(n)


class KernelTimer(object):
    """optional per-launch timing of the HIP-library kernels with events recorded on the stream the
    kernels are launched on (torch's current stream).  Used by bench.py for the `roofline` object:
    it stores (elapsed, algorithmic bytes, flops) per kernel name.  Off by default (TIMER is None)."""

    def __init__(self):
        self.records = {}   # name -> list of (start_event, end_event, bytes, flops)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = {"launches": len(recs), "total_ms": float(sum(ms)), "avg_us": 1e3 * float(sum(ms)) / len(recs),
                         "bytes": float(sum(r[2] for r in recs)), "flops": float(sum(r[3] for r in recs))}
        return out


TIMER = None
FUSED_BN = True    # BatchNorm(+ReLU6)(+residual) through the fused HIP passes; False = ATen F.batch_norm chain


class _timed(object):
    __slots__ = ("name", "nbytes", "flops", "start")

    def __init__(self, name, nbytes=0, flops=0):
        self.name, self.nbytes, self.flops = name, nbytes, flops

    def __enter__(self):
        if TIMER is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()

    def __exit__(self, *exc):
        if TIMER is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            TIMER.records.setdefault(self.name, []).append((self.start, end, self.nbytes, self.flops))
        return False


def _f32_param(w):
    if w.dtype != torch.float32:
        raise _C.OfasrError("weights / filters are fp32 master copies (got %s)" % w.dtype)
    return w


# ------------------------------------------------------------------------------- PixelShuffle
def _shuffle_raw(x, r, inverse):
    _gpu(x)
    x = x.contiguous()
    N, Cx, Hx, Wx = x.shape
    if inverse:
        if Hx % r or Wx % r:
            raise _C.OfasrError("pixel_unshuffle: spatial size %dx%d not divisible by %d" % (Hx, Wx, r))
        C, H, W = Cx, Hx // r, Wx // r
        y = torch.empty((N, C * r * r, H, W), dtype=x.dtype, device=x.device)
        fn = _C.lib().ofasr_pixel_unshuffle
    else:
        if Cx % (r * r):
            raise _C.OfasrError("pixel_shuffle: channels %d not divisible by %d" % (Cx, r * r))
        C, H, W = Cx // (r * r), Hx, Wx
        y = torch.empty((N, C, H * r, W * r), dtype=x.dtype, device=x.device)
        fn = _C.lib().ofasr_pixel_shuffle
    with _timed("pixel_unshuffle" if inverse else "pixel_shuffle", 2 * x.numel() * x.element_size()):
        _C.check(fn(_p(x), _p(y), N, C, H, W, r, x.element_size(), _stream()),
                 "pixel_unshuffle" if inverse else "pixel_shuffle")
    return y


class PixelShuffleFn(Function):
    """nn.PixelShuffle(r) (reference ofa/utils.py:309-310); backward is the inverse permutation."""

    @staticmethod
    def forward(ctx, x, r):
        ctx.r = r
        return _shuffle_raw(x, r, False)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _shuffle_raw(dy, ctx.r, True), None


class PixelUnshuffleFn(Function):
    """pixel_unshuffle (reference ofa/utils.py:383-397, a one-hot strided conv there)."""

    @staticmethod
    def forward(ctx, x, r):
        ctx.r = r
        return _shuffle_raw(x, r, True)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _shuffle_raw(dy, ctx.r, False), None


def pixel_shuffle(x, r=2):
    return PixelShuffleFn.apply(x, r)


def pixel_unshuffle(x, r=2):
    return PixelUnshuffleFn.apply(x, r)


# ------------------------------------------------------------------------- PIL-exact bicubic LR images
def bicubic_resize_u8(img, out_h, out_w):
    """uint8 [..., H, W] planes on the GPU -> [..., out_h, out_w], bit-identical to PIL's
    Image.resize((out_w, out_h), Image.BICUBIC) per plane (ofasr_bicubic_resize_u8, csrc/resample.hip)."""
    _gpu(img)
    if img.dtype != torch.uint8:
        raise _C.OfasrError("bicubic_resize_u8 works on uint8 images (the reference resizes before ToTensor), got %s" % img.dtype)
    img = img.contiguous()
    H, W = img.shape[-2:]
    planes = img.numel() // (H * W)
    out = torch.empty(tuple(img.shape[:-2]) + (out_h, out_w), dtype=torch.uint8, device=img.device)
    L = _C.lib()
    wst, wsp, wsn = _ws(L.ofasr_bicubic_resize_u8_workspace(planes, H, W, out_h, out_w), img.device)
    with _timed("bicubic_resize_u8", img.numel() + out.numel()):
        _C.check(L.ofasr_bicubic_resize_u8(_p(img), _p(out), planes, H, W, out_h, out_w, wsp, wsn, _stream()),
                 "bicubic_resize_u8")
    return out


def lr_images_from_u8(hr_u8):
    """{'image', '2x_down_image', '4x_down_image'} (float32 in [0, 1], the loader contract of the reference's
    Div2K_SetXXDataset.__getitem__, div2k_setxx.py:288-298) from a uint8 HR batch [N, 3, H, W] already on the GPU:
    Scale(1/2) and Scale(1/4) as PIL computes them (output size int(h/f) x int(w/f)), then ToTensor's /255."""
    _, _, H, W = hr_u8.shape
    out = {"image": hr_u8.float().div_(255.0)}
    for f in (2, 4):
        lr = bicubic_resize_u8(hr_u8, int(H * (1.0 / f)), int(W * (1.0 / f)))
        out["%dx_down_image" % f] = lr.float().div_(255.0)
    return out


# --------------------------------------------------------------------------- kernel transform
def _kt_args(chain, mats):
    ks = (ctypes.c_int * len(chain))(*chain)
    n = len(chain) - 1
    arr = (ctypes.c_void_p * max(n, 1))(*([m.data_ptr() for m in mats] if mats else [None] * max(n, 1)))
    return ks, n, arr


class KTransformFn(Function):
    """DynamicSeparableConv2d.get_active_filter (reference dynamic_op.py:46-71).

    forward(w_max [Cmax,1,kmax,kmax], C, chain (kmax, ..., K), transform, *mats) -> f [C,1,K,K]
    `mats[s]` is the '%dto%d_matrix' of chain step s (only when transform and K < kmax).
    backward: dense dw_max (zeros outside rows < C / the crop window) and one gradient per walked
    matrix; matrices of steps not walked are not inputs here, so their .grad stays None."""

    @staticmethod
    def forward(ctx, w_max, C, chain, transform, *mats):
        _gpu(w_max, *mats)
        w = _f32_param(w_max).contiguous()
        mats = [_f32_param(m).contiguous() for m in mats]
        K = chain[-1]
        f = torch.empty((C, 1, K, K), dtype=torch.float32, device=w.device)
        ks, n, arr = _kt_args(chain, mats)
        with _timed("ktransform_fwd", 4 * (C * chain[0] ** 2 + C * K * K + sum(m.numel() for m in mats))):
            _C.check(_C.lib().ofasr_ktransform_fwd(_p(w), ks, n, arr, 1 if transform else 0, _p(f), C, _stream()),
                     "ktransform_fwd")
        ctx.save_for_backward(w, *mats)
        ctx.meta = (C, tuple(chain), bool(transform))
        return f

    @staticmethod
    @once_differentiable
    def backward(ctx, df):
        w, *mats = ctx.saved_tensors
        C, chain, transform = ctx.meta
        df = df.contiguous()
        dw = torch.zeros_like(w)
        dmats = [torch.empty_like(m) for m in mats]
        ks, n, arr = _kt_args(chain, mats)
        darr = (ctypes.c_void_p * max(n, 1))(*([d.data_ptr() for d in dmats] if dmats else [None] * max(n, 1)))
        need = _C.lib().ofasr_ktransform_bwd_workspace(ks, n, C) if (transform and mats) else 0
        wst, wsp, wsn = _ws(need, w.device)
        with _timed("ktransform_bwd", 4 * (2 * C * chain[0] ** 2 + C * chain[-1] ** 2 + 2 * sum(m.numel() for m in mats))):
            _C.check(_C.lib().ofasr_ktransform_bwd(_p(w), ks, n, arr, 1 if transform else 0, _p(df), _p(dw), darr, C,
                                                   wsp, wsn, _stream()), "ktransform_bwd")
        return (dw, None, None, None) + tuple(dmats)


# ------------------------------------------------------------------------------- depthwise conv
class DWConvFn(Function):
    """F.conv2d(x, f, groups=C, padding=k//2) of DynamicSeparableConv2d.forward (dynamic_op.py:79-83)."""

    @staticmethod
    def forward(ctx, x, f):
        _gpu(x, f)
        x = x.contiguous()
        f = _f32_param(f).contiguous()
        N, C, H, W = x.shape
        K = f.shape[-1]
        if f.shape[0] != C:
            raise _C.OfasrError("dwconv: filter has %d channels, input has %d" % (f.shape[0], C))
        y = torch.empty_like(x)
        with _timed("dwconv_fwd_k%d" % K, 2 * x.numel() * x.element_size() + 4 * f.numel(), 2 * K * K * x.numel()):
            _C.check(_C.lib().ofasr_dwconv_fwd(_p(x), _p(f), _p(y), N, C, H, W, K, _dt(x), _stream()), "dwconv_fwd")
        ctx.save_for_backward(x, f)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, f = ctx.saved_tensors
        N, C, H, W = x.shape
        K = f.shape[-1]
        dy = dy.contiguous()
        L = _C.lib()
        dx = df = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with _timed("dwconv_dgrad_k%d" % K, 2 * x.numel() * x.element_size() + 4 * f.numel(), 2 * K * K * x.numel()):
                _C.check(L.ofasr_dwconv_dgrad(_p(dy), _p(f), _p(dx), N, C, H, W, K, _dt(x), _stream()), "dwconv_dgrad")
        if ctx.needs_input_grad[1]:
            df = torch.empty_like(f)
            wst, wsp, wsn = _ws(L.ofasr_dwconv_wgrad_workspace(N, C, H, W, K), x.device)
            with _timed("dwconv_wgrad_k%d" % K, 2 * x.numel() * x.element_size() + 4 * f.numel(), 2 * K * K * x.numel()):
                _C.check(L.ofasr_dwconv_wgrad(_p(dy), _p(x), _p(df), N, C, H, W, K, _dt(x), wsp, wsn, _stream()),
                         "dwconv_wgrad")
        return dx, df


def dwconv(x, f):
    return DWConvFn.apply(x, f)


# ------------------------------------------------------------------------------- pointwise conv
class PWConvFn(Function):
    """DynamicPointConv2d.forward (reference dynamic_op.py:104-112): the [:cout, :cin] slice of the
    max-size 1x1 weight is read IN PLACE (row stride = weight.shape[1]); backward returns a dense
    gradient of the full parameter with exact zeros outside the slice."""

    @staticmethod
    def forward(ctx, x, w_full, cout):
        _gpu(x, w_full)
        x = x.contiguous()
        w = _f32_param(w_full).contiguous()
        if w.dim() != 4 or w.shape[2] != 1 or w.shape[3] != 1:
            raise _C.OfasrError("pwconv: weight must be [Cout_max, Cin_max, 1, 1], got %s" % (tuple(w.shape),))
        N, Cin, H, W = x.shape
        ldw = w.shape[1]
        if Cin > ldw or cout > w.shape[0]:
            raise _C.OfasrError("pwconv: slice [%d,%d] exceeds weight %s" % (cout, Cin, tuple(w.shape)))
        y = torch.empty((N, cout, H, W), dtype=x.dtype, device=x.device)
        with _timed("pwconv_fwd_%dto%d" % (Cin, cout), (x.numel() + y.numel()) * x.element_size() + 4 * Cin * cout,
                    2 * Cin * cout * N * H * W):
            _C.check(_C.lib().ofasr_pwconv_fwd(_p(x), _p(w), ldw, _p(y), N, Cin, cout, H * W, _dt(x), _stream()),
                     "pwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cout = cout
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, Cin, H, W = x.shape
        cout, ldw = ctx.cout, w.shape[1]
        dy = dy.contiguous()
        L = _C.lib()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with _timed("pwconv_dgrad_%dto%d" % (cout, Cin), (x.numel() + dy.numel()) * x.element_size() + 4 * Cin * cout,
                        2 * Cin * cout * N * H * W):
                _C.check(L.ofasr_pwconv_dgrad(_p(dy), _p(w), ldw, _p(dx), N, Cin, cout, H * W, _dt(x), _stream()),
                         "pwconv_dgrad")
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            wst, wsp, wsn = _ws(L.ofasr_pwconv_wgrad_workspace(N, Cin, cout, H * W), x.device)
            with _timed("pwconv_wgrad_%dx%d" % (cout, Cin), (x.numel() + dy.numel()) * x.element_size() + 4 * Cin * cout,
                        2 * Cin * cout * N * H * W):
                _C.check(L.ofasr_pwconv_wgrad(_p(dy), _p(x), _p(dw), ldw, N, Cin, cout, H * W, _dt(x), wsp, wsn,
                                              _stream()), "pwconv_wgrad")
        return dx, dw, None


def pwconv(x, w_full, cout):
    return PWConvFn.apply(x, w_full, cout)


# ------------------------------------------------------------------- BatchNorm + act (+ residual)
ACT_NONE, ACT_RELU6 = 0, 1


class BNActFn(Function):
    """act(BatchNorm2d_[:C](x) (+ residual)) in two HIP passes (statistics, fused apply) and its backward in
    two (reduction, fused apply) -- replaces F.batch_norm on parameter slices (reference dynamic_op.py:148-167)
    + in-place ReLU6 (dynamic_layers.py:44,56) + the shortcut add (proxyless_nets.py:50).

    forward(x, weight, bias, running_mean, running_var, training, momentum, eps, act, residual)
      weight/bias/running_*: the MAX-size parameters / buffers; the first C = x.size(1) entries are used and
      (training) running_* [:C] are updated in place.  Gradients of weight / bias are dense max-size
      tensors with zeros beyond C (what autograd of the slice gives in the reference)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, residual):
        _gpu(x, weight, bias, running_mean, running_var, residual)
        x = x.contiguous()
        N, C, H, W = x.shape
        HW = H * W
        L = _C.lib()
        dev = x.device
        w = _f32_param(weight)
        b = _f32_param(bias)
        stats = torch.empty((4, C), dtype=torch.float32, device=dev)   # mean, invstd, scale, shift
        if residual is not None:
            residual = residual.contiguous()
            if residual.shape != x.shape or residual.dtype != x.dtype:
                raise _C.OfasrError("bn_act: residual %s/%s does not match x %s/%s" % (
                    tuple(residual.shape), residual.dtype, tuple(x.shape), x.dtype))
        y = torch.empty_like(x)
        wst, wsp, wsn = _ws(L.ofasr_bn_workspace(N, C) if training else 0, dev)
        rm = _p(running_mean) if running_mean is not None else ctypes.c_void_p(None)
        rv = _p(running_var) if running_var is not None else ctypes.c_void_p(None)
        rp = _p(residual) if residual is not None else ctypes.c_void_p(None)
        with _timed("bn_fwd", (3 + (residual is not None)) * x.numel() * x.element_size()):
            _C.check(L.ofasr_bn_fwd(_p(x), rp, _p(y), _p(w), _p(b), rm, rv, float(momentum), float(eps),
                                    1 if training else 0, _p(stats), N, C, HW, act, _dt(x), wsp, wsn, _stream()),
                     "bn_fwd")
        keep_res = residual if (residual is not None and act != ACT_NONE) else None
        ctx.save_for_backward(x, stats, keep_res)
        ctx.meta = (bool(training), act, residual is not None, tuple(weight.shape))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, stats, res = ctx.saved_tensors
        dx, dgamma, dbeta, dres = _bn_act_backward(x, stats, res, ctx.meta, dy)
        return dx, dgamma, dbeta, None, None, None, None, None, None, dres


def _bn_act_backward(x, stats, res, meta, dy):
    """(dx, dgamma, dbeta, dresidual) of act(BN(x) (+ residual)): reduction + fused apply (ofasr_bn_act_bwd)"""
    training, act, has_res, wshape = meta
    N, C, H, W = x.shape
    HW = H * W
    L = _C.lib()
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    # the kernel writes channels [0, C); only a sliced BN (C < num_features) needs the zero tail
    full = int(wshape[0]) == C
    gb = (torch.empty if full else torch.zeros)((2,) + tuple(wshape), dtype=torch.float32, device=x.device)
    dgamma, dbeta = gb[0], gb[1]
    dres = None
    if has_res:
        # without an activation the residual gradient IS dy; with one it is the masked dy
        dres = torch.empty_like(x) if act != ACT_NONE else dy
    wst, wsp, wsn = _ws(L.ofasr_bn_act_bwd_workspace(N, C), x.device)
    rp = _p(res) if res is not None else ctypes.c_void_p(None)
    drp = _p(dres) if (has_res and act != ACT_NONE) else ctypes.c_void_p(None)
    with _timed("bn_act_bwd", 5 * x.numel() * x.element_size()):
        _C.check(L.ofasr_bn_act_bwd(_p(dy), _p(x), rp, _p(dx), drp, _p(stats[2]), _p(stats[3]), _p(stats[0]),
                                    _p(stats[1]), _p(dgamma), _p(dbeta), N, C, HW, act, 1 if training else 0,
                                    _dt(x), wsp, wsn, _stream()), "bn_act_bwd")
    return dx, dgamma, dbeta, dres


class BNActCPFn(Function):
    """BatchNorm (+ReLU6 | + PixelShuffle(2)) of a conv output whose batch statistics the conv's epilogue already took
    (Conv2dStatFn: [C][units] (sum, sum of squares) partials) -- ConvLayer in training mode, reference ofa/layers.py:
    120-151.  act: ACT_NONE / ACT_RELU6 -> ofasr_bn_fwd_cp (fold + apply in one kernel); ACT_PIXEL_SHUFFLE2 ->
    ofasr_bn_finalize_cp + ofasr_pixel_shuffle2_bn (the BN apply writes the up-sampled tensor: no shuffle pass).
    Backward: the shuffle's inverse on the incoming gradient, then BNActFn's backward."""

    @staticmethod
    def forward(ctx, x, partial, weight, bias, running_mean, running_var, momentum, eps, act):
        _gpu(x, weight, bias, running_mean, running_var)
        N, C, H, W = x.shape
        L = _C.lib()
        w, b = _f32_param(weight), _f32_param(bias)
        stats = torch.empty((4, C), dtype=torch.float32, device=x.device)
        units = partial.shape[1]
        if act == ACT_PIXEL_SHUFFLE2:
            y = torch.empty((N, C // 4, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
            _C.check(L.ofasr_bn_finalize_cp(_p(partial), units, C, float(N * H * W), _p(w), _p(b), _p(running_mean),
                                            _p(running_var), float(momentum), float(eps), 1, _p(stats), _stream()),
                     "bn_finalize_cp")
            with _timed("bn_pixel_shuffle", 2 * x.numel() * x.element_size()):
                _C.check(L.ofasr_pixel_shuffle2_bn(_p(x), _p(y), _p(stats), N, C // 4, H, W, _dt(x), _stream()),
                         "pixel_shuffle2_bn")
        else:
            y = torch.empty_like(x)
            with _timed("bn_fwd_cp", 2 * x.numel() * x.element_size()):
                _C.check(L.ofasr_bn_fwd_cp(_p(x), None, _p(y), _p(partial), units, _p(w), _p(b), _p(running_mean),
                                           _p(running_var), float(momentum), float(eps), 1, _p(stats), N, C, H * W, int(act),
                                           _dt(x), _stream()), "bn_fwd_cp")
        ctx.save_for_backward(x, stats)
        ctx.meta = (True, ACT_NONE if act == ACT_PIXEL_SHUFFLE2 else act, False, tuple(weight.shape))
        ctx.shuffled = act == ACT_PIXEL_SHUFFLE2
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        if ctx.shuffled and PS_BWD_FUSED:
            # both BatchNorm-backward passes read the gradient through the inverse shuffle (no un-shuffle pass)
            N, C, H, W = x.shape
            L = _C.lib()
            dy = dy.contiguous()
            dx = torch.empty_like(x)
            wshape = ctx.meta[3]
            full = int(wshape[0]) == C
            gb = (torch.empty if full else torch.zeros)((2,) + tuple(wshape), dtype=torch.float32, device=x.device)
            wst, wsp, wsn = _ws(L.ofasr_bn_bwd_ps2_workspace(N, C), x.device)
            with _timed("bn_bwd_ps2", 5 * x.numel() * x.element_size()):
                _C.check(L.ofasr_bn_bwd_ps2(_p(dy), _p(x), _p(dx), _p(stats[2]), _p(stats[0]), _p(stats[1]), _p(gb[0]),
                                            _p(gb[1]), N, C, H, W, 1, _dt(x), wsp, wsn, _stream()), "bn_bwd_ps2")
            return dx, None, gb[0], gb[1], None, None, None, None, None
        if ctx.shuffled:
            dy = _shuffle_raw(dy.contiguous(), 2, True)
        dx, dgamma, dbeta, _ = _bn_act_backward(x, stats, None, ctx.meta, dy)
        return dx, None, dgamma, dbeta, None, None, None, None, None


def bn_act(x, bn, act=ACT_NONE, residual=None):
    """apply nn.BatchNorm2d `bn` (its first x.size(1) channels) + activation (+ residual) through the fused
    HIP kernels, with nn.BatchNorm2d's bookkeeping (num_batches_tracked, momentum=None -> cumulative average)."""
    training = bn.training or not bn.track_running_stats
    factor = 0.0
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    return BNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, factor, bn.eps, act,
                         residual)


# ------------------------------------------------------------------------------ fused MB block
FUSED_BLOCK = True   # DynamicMBConvLayer (+ shortcut) through ONE composite FFI call per direction


# Deferred weight gradients of the composite block.  ofasr_mbconv_bwd computes the weight / transform-matrix gradients
# on the library's side stream; joining that stream at the end of every block's backward makes the next block's
# input-gradient chain queue behind them (7 % of the north-star step).  In deferred mode the backward node returns dx
# and the BN gradients only and keeps the call's buffers alive; flush_deferred() then joins the side stream ONCE and does
# what AccumulateGrad would have done (p.grad = g, or p.grad += g).
#
# Who calls flush_deferred():
#   * the trainers of this package right after loss.backward() (SRRunManager.train_one_epoch,
#     progressive_shrinking.train_one_epoch, bench.py), and FlatGradReducer.reduce();
#   * safety nets, all public torch API: a global optimizer-step pre-hook (so `loss.backward(); optimizer.step()` --
#     the reference's loop, progressive_shrinking.py:199-203 -- needs no change), the next composite forward, and
#     Module.zero_grad through the same pre-hook path of the next step.
#   Code that READS `.grad` between backward() and optimizer.step() (gradient clipping, logging) calls
#   ops.flush_deferred() first, or runs with deferred_weight_grads(False) / OFASR_MBCONV_DEFER_JOIN=0.
#   OFASR_DEFER_ENGINE_CALLBACK=1 additionally installs the flush as an end-of-backward callback of the autograd engine
#   (torch.autograd.Variable._execution_engine.queue_callback, the private hook DistributedDataParallel uses): `.grad` is
#   then complete when backward() returns.  Off by default because it is not public torch API.
# Visible differences in deferred mode: those .grad fields appear at the flush instead of mid-way through backward();
# tensor hooks and torch's post-accumulate-grad hooks registered on the deferred weights do not run for them (use
# register_deferred_grad_hook below -- FlatGradReducer does); torch.autograd.grad() does not return them.  Only leaf
# parameters are deferred.
DEFER_WGRAD = os.environ.get("OFASR_MBCONV_DEFER_JOIN", "1") != "0"
DEFER_ENGINE_CALLBACK = os.environ.get("OFASR_DEFER_ENGINE_CALLBACK", "0") != "0"


class _Deferred(object):
    queued = False      # a flush callback is installed for the running backward pass
    lib_mode = None     # last value handed to ofasr_mbconv_defer_join
    keep = []           # buffers the side stream may still be using
    grads = []          # (parameter, gradient) pairs to accumulate after the join
    ext = {}            # device index -> torch view of the library's side stream (None when it is disabled)
    ext_used = set()    # devices whose side stream got work from this module since the last flush
    late = []           # launches of deferred static-conv weight gradients held back until the MB stack's backward starts
    hooks = {}          # id(parameter) -> [callables]: run after a deferred gradient has been accumulated
    opt_hook = None     # handle of the global optimizer-step pre-hook


def _lib_side_stream(device):
    """the library's side stream as a torch stream (so that torch events / `with torch.cuda.stream` work on it)."""
    key = torch.device(device).index or 0
    if key not in _Deferred.ext:
        h = _C.lib().ofasr_side_stream()
        _Deferred.ext[key] = torch.cuda.ExternalStream(h, device=device) if h else None
    return _Deferred.ext[key]


def deferred_weight_grads(enable=True):
    """switch the deferred mode (see above); returns the previous setting."""
    global DEFER_WGRAD
    was, DEFER_WGRAD = DEFER_WGRAD, bool(enable)
    return was


def _set_lib_defer(on):
    if _Deferred.lib_mode is not on:
        _C.lib().ofasr_mbconv_defer_join(1 if on else 0)
        _Deferred.lib_mode = on


def register_deferred_grad_hook(param, fn):
    """fn(param) runs after flush_deferred() has put a deferred gradient into param.grad -- the public stand-in for
    torch's post-accumulate-grad hooks, which autograd only runs for gradients IT accumulates.  Returns a remover."""
    lst = _Deferred.hooks.setdefault(id(param), [])
    lst.append(fn)

    def remove():
        if fn in lst:
            lst.remove(fn)
    return remove


def launch_late_conv_wgrads():
    """start the static-conv weight gradients that were held back (CONV_WGRAD_LATE) on the library's side stream"""
    if _Deferred.late:
        jobs, _Deferred.late = _Deferred.late, []
        for job in jobs:
            job()


def flush_deferred():
    """join the library's side stream into the current stream, then accumulate the held weight gradients."""
    _Deferred.queued = False
    launch_late_conv_wgrads()
    if not (_Deferred.keep or _Deferred.grads):
        return
    _C.check(_C.lib().ofasr_mbconv_join(_stream()), "mbconv_join")
    for key in _Deferred.ext_used:      # work this module put on the side stream itself (static-conv weight gradients)
        torch.cuda.current_stream(key).wait_stream(_Deferred.ext[key])
    _Deferred.ext_used = set()
    grads, _Deferred.grads, _Deferred.keep = _Deferred.grads, [], []
    with torch.no_grad():
        acc_dst, acc_src, seen = [], [], set()
        for p, g in grads:
            if p.grad is None:
                p.grad = g
            elif id(p) in seen:        # a parameter used by two blocks: keep its adds ordered
                p.grad.add_(g)
            else:                      # gradient accumulation / the data-parallel bucket's views
                seen.add(id(p))
                acc_dst.append(p.grad)
                acc_src.append(g)
        if acc_dst:                    # one multi-tensor launch instead of one add per parameter
            torch._foreach_add_(acc_dst, acc_src)
        if _Deferred.hooks:
            for p, _ in grads:
                for h in list(_Deferred.hooks.get(id(p), ())):
                    h(p)


_flush_deferred = flush_deferred   # (round-2 name)


def single_thread_backward(enable=True):
    """run backward() on the calling thread (torch.autograd.set_multithreading_enabled, public API).  The autograd engine
    otherwise hands every backward pass to a per-device worker thread; with one process per GPU there is nothing for
    that thread to overlap with, and the hand-off costs ~1.4 ms of host time per training step here (3.0 against 1.66 ms
    of backward host time, tools/host_profile.py) -- more than a fifth of the step.  The trainers of this package and
    bench.py call this once; it is process-wide."""
    torch.autograd.set_multithreading_enabled(not enable)


def _install_flush_nets():
    """the optimizer-step safety net: any torch optimizer flushes before it reads gradients (public API)"""
    if _Deferred.opt_hook is None:
        from torch.optim.optimizer import register_optimizer_step_pre_hook
        _Deferred.opt_hook = register_optimizer_step_pre_hook(lambda opt, args, kwargs: flush_deferred())


# scratch of the composite backward: a fresh tensor per call, or (SHARED_TMP) one cached buffer per size that every
# block reuses -- the library orders a call behind unjoined side work that still reads the same bytes
SHARED_TMP = os.environ.get("OFASR_MBCONV_SHARED_TMP", "0") != "0"
_TMP_CACHE = {}


def _bwd_scratch(numel, dtype, device):
    if not SHARED_TMP:
        return torch.empty(numel, dtype=dtype, device=device)
    key = (numel, dtype, str(device))
    t = _TMP_CACHE.get(key)
    if t is None:
        t = _TMP_CACHE[key] = torch.empty(numel, dtype=dtype, device=device)
    return t


def _defer_this_backward(params):
    """True when this composite backward may leave its weight gradients to flush_deferred()."""
    if not DEFER_WGRAD or not all(p.is_leaf for p in params):
        return False
    _install_flush_nets()
    if DEFER_ENGINE_CALLBACK and not _Deferred.queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred)
            _Deferred.queued = True
        except (RuntimeError, AttributeError):      # not inside an engine-driven backward pass / API gone
            pass
    return True


def _mbconv_desc(x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, mats):
    """ofasr_mbconv_desc of one block call (include/ofasr.h) from the host mirror's tensors"""
    N, Cin, H, W = x.shape
    d = _C.MBConvDesc()
    d.N, d.Cin, d.mid, d.Cout, d.H, d.W = N, Cin, cfg["mid"], cfg["out"], H, W
    d.K = cfg["K"]
    for i, k in enumerate(cfg["chain"]):
        d.ks[i] = k
    d.chain_len = len(cfg["chain"])
    d.transform = 1 if mats else 0
    d.dtype = _dt(x)
    d.residual = 1 if cfg["residual"] else 0
    gam, bet = (g1, g2, g3), (b1, b2, b3)
    for i, bn in enumerate(cfg["bns"]):
        training = bn.training or not bn.track_running_stats
        d.bn_training[i] = 1 if training else 0
        upd = bn.training and bn.track_running_stats
        d.bn_momentum[i] = float(bn.momentum) if upd else 0.0
        d.bn_eps[i] = float(bn.eps)
        d.gamma[i] = gam[i].data_ptr()
        d.beta[i] = bet[i].data_ptr()
        d.running_mean[i] = bn.running_mean.data_ptr()
        d.running_var[i] = bn.running_var.data_ptr()
        d.num_batches_tracked[i] = bn.num_batches_tracked.data_ptr() if (upd and bn.num_batches_tracked is not None) else None
    d.Cmid_max, d.Cout_max = w1.shape[0], w2.shape[0]
    d.ldw1, d.ldw2 = w1.shape[1], w2.shape[1]
    d.w1, d.w2, d.wdw_max = w1.data_ptr(), w2.data_ptr(), wdw.data_ptr()
    for i, m in enumerate(mats):
        d.mats[i] = m.data_ptr()
    return d


FUSED_INFER = os.environ.get("OFASR_MBCONV_FUSED_INFER", "1") != "0"   # the one-kernel eval-mode block (ofasr_mbconv_infer)


# Inference operands (BN-folded weight images of the one-kernel MB block, weight images + BN scale/shift of the static
# convs) are a function of the parameters only; in eval mode they are prepared once and kept while the tensors they were
# built from are the SAME Python objects (weak references: an entry dies with its tensors, so a new tensor that the
# allocator places at an old address never matches) at the same address and version (torch bumps Tensor._version on
# every tracked in-place write: optimizer steps, load_state_dict, copy_).  Writes through `.data` are invisible to the
# version counter, so everything in this package that rewrites weights that way (re_organize_middle_weights, init_model,
# copy_bn) and every switch back to training mode calls clear_infer_cache().  OFASR_INFER_OPERAND_CACHE=0 prepares
# per call.
INFER_CACHE = os.environ.get("OFASR_INFER_OPERAND_CACHE", "1") != "0"
_INFER_OPERANDS = collections.OrderedDict()
_INFER_CACHE_MAX = 256


_INFER_EPOCH = [0]


def clear_infer_cache():
    _INFER_OPERANDS.clear()
    _INFER_EPOCH[0] += 1


def infer_epoch():
    """number of clear_infer_cache() calls so far (graphed.GraphedEval keys its captured graphs on it)"""
    return _INFER_EPOCH[0]


def infer_operand_buffers():
    """the operand buffers currently cached (a captured graph that points at them keeps them alive)"""
    return [v[0] for v in _INFER_OPERANDS.values()]


def _infer_operands(kind, tensors, nbytes, device, prepare):
    """the prepared operand buffer of (kind, tensors); prepare(ptr, nbytes) fills a new one on a miss"""
    if not INFER_CACHE:
        buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        prepare(ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(buf.numel()))
        return buf
    key = (kind, str(device)) + tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors)
    hit = _INFER_OPERANDS.get(key)
    if hit is not None:
        buf, refs = hit
        if all((r is None and t is None) or (r is not None and r() is t) for r, t in zip(refs, tensors)):
            _INFER_OPERANDS.move_to_end(key)
            return buf
    buf = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    prepare(ctypes.c_void_p(buf.data_ptr()), ctypes.c_size_t(buf.numel()))
    _INFER_OPERANDS[key] = (buf, [weakref.ref(t) if t is not None else None for t in tensors])
    while len(_INFER_OPERANDS) > _INFER_CACHE_MAX:
        _INFER_OPERANDS.popitem(last=False)
    return buf


def mbconv_infer(x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, *mats):
    """the whole MB block (+ shortcut) as ONE kernel, forward only, eval-mode BN folded into the convolutions
    (ofasr_mbconv_infer_prepare / _run, csrc/mbfused.hip).  Returns None when the shape / dtype / BN mode is outside
    what the kernel implements (the caller then takes the composite path)."""
    _gpu(x, w1, wdw, w2)
    if x.dtype not in (torch.float16, torch.bfloat16):
        return None
    x = x.contiguous()
    L = _C.lib()
    d = _mbconv_desc(x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, mats)
    dp = ctypes.byref(d)
    if not L.ofasr_mbconv_infer_supported(dp):
        return None
    out = torch.empty((x.shape[0], cfg["out"], x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device)
    bns = cfg["bns"]
    kind = ("mb", x.dtype, cfg["mid"], cfg["out"], cfg["K"], tuple(cfg["chain"]), int(d.transform),
            tuple(float(bn.eps) for bn in bns))
    tensors = (w1, g1, b1, wdw, g2, b2, w2, g3, b3) + tuple(mats) + tuple(t for bn in bns
                                                                         for t in (bn.running_mean, bn.running_var))

    def prepare(ptr, nbytes):
        _C.check(L.ofasr_mbconv_infer_prepare(dp, ptr, nbytes, _stream()), "mbconv_infer_prepare")

    with _timed("mbconv_infer"):
        opnd = _infer_operands(kind, tensors, L.ofasr_mbconv_infer_operand_bytes(dp), x.device, prepare)
        sst, ssp, ssn = _ws(L.ofasr_mbconv_infer_scratch_bytes(dp), x.device)
        _C.check(L.ofasr_mbconv_infer_run(dp, _p(x), _p(out), _p(opnd), ctypes.c_size_t(opnd.numel()), ssp, ssn, _stream()),
                 "mbconv_infer_run")
    return out


class FusedMBConvFn(Function):
    """DynamicMBConvLayer.forward (+ identity shortcut) as ONE host call per direction (ofasr_mbconv_fwd/_bwd,
    include/ofasr.h): expand 1x1 -> BN+ReLU6 -> kernel transform -> depthwise -> BN+ReLU6 -> project 1x1 -> BN (+x).
    Same kernels as the per-op Functions above; what changes is the host cost (1 autograd node and 1 FFI call
    instead of 7 nodes / ~25 calls per block), which bounds the training step at the MB stack's sizes.

    apply(x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, *mats); cfg = dict built by DynamicMBConvLayer.forward."""

    @staticmethod
    def forward(ctx, x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, *mats):
        _gpu(x, w1, wdw, w2)
        if _Deferred.grads or _Deferred.keep:   # a backward pass that died before its callback: settle it now
            _flush_deferred()
        x = x.contiguous()
        L = _C.lib()
        N, Cin, H, W = x.shape
        mid, Cout = cfg["mid"], cfg["out"]
        bns = cfg["bns"]
        d = _mbconv_desc(x, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, mats)
        dp = ctypes.byref(d)
        HW = H * W
        act = torch.empty(L.ofasr_mbconv_act_elems(dp), dtype=x.dtype, device=x.device)   # y1 | y2 | y3 | out when fused
        stat = torch.empty(L.ofasr_mbconv_stat_floats(dp), dtype=torch.float32, device=x.device)
        wsn = L.ofasr_mbconv_workspace(dp)
        ws = torch.empty(wsn, dtype=torch.uint8, device=x.device)
        with _timed("mbconv_fwd"):
            _C.check(L.ofasr_mbconv_fwd(dp, _p(x), _p(act), _p(stat), _p(ws), wsn, _stream()), "mbconv_fwd")
        ctx.save_for_backward(x, act, stat, w1, wdw, w2, g1, g2, g3, *mats)
        ctx.desc = d
        ctx.keep = (bns, ws)       # keeps the BN buffers the descriptor points at alive; workspace reused in backward
        return act[act.numel() - N * HW * Cout:].view(N, Cout, H, W)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, act, stat, w1, wdw, w2, g1, g2, g3, *mats = ctx.saved_tensors
        d = ctx.desc
        bns, ws = ctx.keep
        L = _C.lib()
        N, Cin, H, W = x.shape
        HW = H * W
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        tmp = _bwd_scratch(N * HW * (3 * d.mid + d.Cout), x.dtype, x.device)
        # the nine zero-initialised buffers first and adjacent: the library then clears them with one fill
        sizes = [w1.numel(), w2.numel(), wdw.numel()] + [g1.numel()] * 2 + [g2.numel()] * 2 + [g3.numel()] * 2 \
            + [m.numel() for m in mats]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=x.device)
        parts = flat.split(sizes)
        dw1, dw2, dwdw = parts[0].view_as(w1), parts[1].view_as(w2), parts[2].view_as(wdw)
        dg1, db1, dg2, db2, dg3, db3 = parts[3:9]
        dmats = [parts[9 + i].view_as(mats[i]) for i in range(len(mats))]
        g = _C.MBConvGrads()
        g.dw1, g.dw2, g.dwdw_max = dw1.data_ptr(), dw2.data_ptr(), dwdw.data_ptr()
        for i, m in enumerate(dmats):
            g.dmats[i] = m.data_ptr()
        for i, (a, b) in enumerate(((dg1, db1), (dg2, db2), (dg3, db3))):
            g.dgamma[i], g.dbeta[i] = a.data_ptr(), b.data_ptr()
        defer = _defer_this_backward((w1, wdw, w2) + tuple(mats))
        _set_lib_defer(defer)
        with _timed("mbconv_bwd"):
            _C.check(L.ofasr_mbconv_bwd(ctypes.byref(d), _p(x), _p(act), _p(stat), _p(dout), _p(dx), _p(tmp),
                                        ctypes.byref(g), _p(ws), ws.numel(), _stream()), "mbconv_bwd")
        if defer:
            _Deferred.keep.append((x, act, stat, dout, tmp, ws, flat, bns, d, g, w1, wdw, w2, mats))
            need = ctx.needs_input_grad
            pairs = [(w1, dw1, need[2]), (wdw, dwdw, need[5]), (w2, dw2, need[8])] + \
                [(m, dm, need[11 + i]) for i, (m, dm) in enumerate(zip(mats, dmats))]
            _Deferred.grads.extend((p, gr) for p, gr, wanted in pairs if wanted)
            return (dx, None, None, dg1, db1, None, dg2, db2, None, dg3, db3) + (None,) * len(dmats)
        return (dx, None, dw1, dg1, db1, dwdw, dg2, db2, dw2, dg3, db3) + tuple(dmats)


# ------------------------------------------------------------------------ backward milestones
# A point of the network that, once its backward node runs, tells a registered listener that every gradient computed
# "after" it in forward order has been issued (the data-parallel reducer starts exchanging the decoder tail's gradients
# there, while the MB stack's backward still runs).  The node is only inserted while a listener is registered.
_MILESTONES = {}


def register_grad_milestone(tag, fn):
    """fn() is called from backward when the pass reaches grad_milestone(x, tag); returns a remover"""
    _MILESTONES[tag] = fn

    def remove():
        if _MILESTONES.get(tag) is fn:
            del _MILESTONES[tag]
    return remove


class _MilestoneFn(Function):
    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        fn = _MILESTONES.get(ctx.tag)
        if fn is not None:
            fn()
        return g, None


def grad_milestone(x, tag):
    if tag not in _MILESTONES or not torch.is_grad_enabled() or not x.requires_grad:
        return x
    return _MilestoneFn.apply(x, tag)


# ------------------------------------------------------------------------------------ MB stack
# All active MB blocks of the network in ONE autograd node and ONE foreign call per direction (ofasr_mbstack_fwd / _bwd).
# Per block and direction the per-block path costs ~100 us of host time above the C ABI (module dispatch, autograd node,
# descriptor marshalling, three allocations) -- 4.6 ms per step against 6.4 ms of GPU time in round 2; here the
# descriptors are cached per (block, sub-network, shape) and the buffers of all blocks are slices of three allocations.
FUSED_STACK = os.environ.get("OFASR_MBSTACK", "1") != "0"
_DESC_CACHE = {}
_PLAN_CACHE = {}
_ALIGN = 256


def _cached_desc(x_shape, dtype, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, mats):
    """(descriptor, act elements, stat floats, workspace bytes) of one block call.  Everything in the descriptor is
    covered by the key: shapes, sub-network, BN modes / momentum / eps, and the addresses the C side dereferences."""
    bns = cfg["bns"]
    key = (id(cfg["owner"]), x_shape, dtype, cfg["mid"], cfg["out"], cfg["K"], cfg["residual"], len(mats),
           w1.data_ptr(), wdw.data_ptr(), w2.data_ptr(), g1.data_ptr(), g2.data_ptr(), g3.data_ptr(),
           tuple(m.data_ptr() for m in mats),
           tuple((bn.training, bn.track_running_stats, bn.momentum, bn.eps, bn.running_mean.data_ptr(),
                  bn.running_var.data_ptr()) for bn in bns))
    hit = _DESC_CACHE.get(key)
    if hit is None:
        class _X(object):   # what _mbconv_desc reads of x
            pass
        fake = _X()
        fake.shape, fake.dtype = x_shape, dtype
        d = _mbconv_desc(fake, cfg, w1, g1, b1, wdw, g2, b2, w2, g3, b3, mats)
        L = _C.lib()
        dp = ctypes.byref(d)
        hit = (d, int(L.ofasr_mbconv_act_elems(dp)), int(L.ofasr_mbconv_stat_floats(dp)), int(L.ofasr_mbconv_workspace(dp)))
        if len(_DESC_CACHE) > 4096:
            _DESC_CACHE.clear()
        _DESC_CACHE[key] = hit
    return hit


def _round_up(n, a=_ALIGN):
    return (n + a - 1) // a * a


class FusedMBStackFn(Function):
    """the stage loop of OFAMobileNetS4.forward (reference ofa_mbs4.py:147-151: every active MobileInvertedResidualBlock
    in turn) as one node.  apply(x, blocks, *params): blocks = [cfg, ...] as DynamicMBConvLayer.composite_args() builds
    them, each with cfg["nparams"] tensors in params (w1, g1, b1, wdw, g2, b2, w2, g3, b3, *mats)."""

    @staticmethod
    def forward(ctx, x, blocks, *params):
        _gpu(x)
        if _Deferred.grads or _Deferred.keep:   # a backward pass whose gradients were never collected: settle it now
            flush_deferred()
        x = x.contiguous()
        L = _C.lib()
        N, C, H, W = x.shape
        es = x.element_size()
        n = len(blocks)
        items = (_C.MBStackItem * n)()
        metas, off, shape = [], 0, (N, C, H, W)
        act_b = stat_b = ws_b = 0
        for i, cfg in enumerate(blocks):
            ps = params[off:off + cfg["nparams"]]
            off += cfg["nparams"]
            d, act_n, stat_n, ws_n = _cached_desc(shape, x.dtype, cfg, *ps[:9], ps[9:])
            metas.append((d, act_b, act_n, stat_b, stat_n, ws_b, ws_n, ps))
            act_b += _round_up(act_n * es)
            stat_b += _round_up(stat_n * 4)
            ws_b += _round_up(ws_n)
            shape = (N, cfg["out"], H, W)
        pool = torch.empty(act_b + stat_b + ws_b, dtype=torch.uint8, device=x.device)
        base = pool.data_ptr()
        for i, (d, a0, act_n, s0, stat_n, w0, ws_n, ps) in enumerate(metas):
            it = items[i]
            it.desc = ctypes.addressof(d)
            it.act_buf = base + a0
            it.stat_buf = base + act_b + s0
            it.workspace = base + act_b + stat_b + w0
            it.workspace_bytes = ws_n
        with _timed("mbstack_fwd"):
            _C.check(L.ofasr_mbstack_fwd(items, n, _p(x), _stream()), "mbstack_fwd")
        d, a0, act_n = metas[-1][0], metas[-1][1], metas[-1][2]
        out_n = N * blocks[-1]["out"] * H * W
        out = pool[a0 + (act_n - out_n) * es:a0 + act_n * es].view(x.dtype).view(N, blocks[-1]["out"], H, W)
        ctx.save_for_backward(x, pool, *params)
        ctx.stack = (items, metas, [cfg["bns"] for cfg in blocks], (act_b, stat_b, ws_b))
        return out

    @staticmethod
    def _plan(metas, N, H, W, es):
        """layout of the backward buffers (a function of the descriptors only; cached with them): per block the offset of
        its tmp scratch, and the flat fp32 gradient buffer [w1 | w2 | wdw | dg1 | db1 | dg2 | db2 | dg3 | db3 | mats...]"""
        key = (N, H, W, es) + tuple((id(m[0]), m[0].mid, m[0].Cout, len(m[7])) for m in metas)
        hit = _PLAN_CACHE.get(key)
        if hit is None:
            tmp_off, tmp_b, sizes, shapes, g_off, g_n = [], 0, [], [], [], 0
            for (d, a0, act_n, s0, stat_n, w0, ws_n, ps) in metas:
                tmp_off.append(tmp_b)
                tmp_b += _round_up(N * H * W * (3 * d.mid + d.Cout) * es)
                w1, g1, b1, wdw, g2, b2, w2, g3, b3 = ps[:9]
                sz = [w1.numel(), w2.numel(), wdw.numel()] + [g1.numel()] * 2 + [g2.numel()] * 2 + [g3.numel()] * 2 + \
                    [m.numel() for m in ps[9:]]
                offs, o = [], g_n
                for n_ in sz:
                    offs.append(4 * o)
                    o += n_
                g_off.append(offs)
                g_n = o
                sizes.extend(sz)
                shapes.append((tuple(w1.shape), tuple(w2.shape), tuple(wdw.shape), [tuple(m.shape) for m in ps[9:]]))
            hit = (tmp_off, tmp_b, sizes, shapes, g_off, g_n)
            if len(_PLAN_CACHE) > 1024:
                _PLAN_CACHE.clear()
            _PLAN_CACHE[key] = hit
        return hit

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        x, pool, *params = ctx.saved_tensors
        items, metas, bns, _ = ctx.stack
        L = _C.lib()
        N, C, H, W = x.shape
        n = len(metas)
        dout = dout.contiguous()
        launch_late_conv_wgrads()    # the decoder tail's weight gradients: beside this stack's bandwidth-bound chain
        tmp_off, tmp_b, sizes, shapes, g_off, g_n = FusedMBStackFn._plan(metas, N, H, W, x.element_size())
        # per block: tmp scratch, the dense gradient buffers (ONE fp32 allocation for the whole stack, cleared by the
        # library with one fill), and dx -- two alternating buffers (dx is only read on this stream) plus the stack's own
        tmp = _bwd_scratch(tmp_b, torch.uint8, x.device)
        flat = torch.empty(g_n, dtype=torch.float32, device=x.device)
        parts = flat.split(sizes)
        dx = torch.empty_like(x)
        pp = [torch.empty_like(x), torch.empty_like(x)] if n > 1 else []
        all_leaf = all(p.is_leaf for (_, _, _, _, _, _, _, ps) in metas for p in (ps[0], ps[3], ps[6]) + tuple(ps[9:]))
        defer = all_leaf and _defer_this_backward(())
        _set_lib_defer(defer)
        need = ctx.needs_input_grad
        fb, tb = flat.data_ptr(), tmp.data_ptr()
        dxp = [dx.data_ptr()] + [pp[i & 1].data_ptr() for i in range(1, n)]
        gstructs, grads_out, deferred_pairs = [], [None, None], []
        pi, k = 2, 0
        for i, (d, a0, act_n, s0, stat_n, w0, ws_n, ps) in enumerate(metas):
            offs = g_off[i]
            nm = len(ps) - 9
            g = _C.MBConvGrads()
            g.dw1, g.dw2, g.dwdw_max = fb + offs[0], fb + offs[1], fb + offs[2]
            for j in range(nm):
                g.dmats[j] = fb + offs[9 + j]
            for j in range(3):
                g.dgamma[j], g.dbeta[j] = fb + offs[3 + 2 * j], fb + offs[4 + 2 * j]
            gstructs.append(g)
            it = items[i]
            it.tmp_buf = tb + tmp_off[i]
            it.grads = ctypes.addressof(g)
            it.dx = dxp[i]
            pt = parts[k:k + 9 + nm]
            k += 9 + nm
            sh = shapes[i]
            # gradients in input order: w1, g1, b1, wdw, g2, b2, w2, g3, b3, *mats (weights and matrices may be deferred)
            order = ((0, sh[0]), (3, None), (4, None), (2, sh[2]), (5, None), (6, None), (1, sh[1]), (7, None), (8, None)) + \
                tuple((9 + j, sh[3][j]) for j in range(nm))
            for j, (src, shape) in enumerate(order):
                if shape is None:
                    grads_out.append(pt[src])
                elif defer:
                    if need[pi + j]:
                        deferred_pairs.append((ps[j], pt[src].view(shape)))
                    grads_out.append(None)
                else:
                    grads_out.append(pt[src].view(shape))
            pi += len(ps)
        with _timed("mbstack_bwd"):
            _C.check(L.ofasr_mbstack_bwd(items, n, _p(x), _p(dout), _stream()), "mbstack_bwd")
        if defer:
            _Deferred.keep.append((x, pool, dout, tmp, flat, pp, dx, bns, items, metas, gstructs, params))
            _Deferred.grads.extend(deferred_pairs)
        return tuple([dx] + grads_out[1:])


def mbstack(x, blocks, params):
    return FusedMBStackFn.apply(x, blocks, *params)


# ------------------------------------------------------------------------------- dense KxK conv
# static conv backward: weight gradient on a side stream beside the input gradient.  Off by default: measured 4 %
# slower on the north-star step (tools/ab_side.py) -- the two MFMA kernels each want every CU's LDS -- while the same
# fork/join inside the composite MB-block backward (bandwidth-bound kernels, csrc/mbconv.hip) is 6 % faster.
SIDE_STREAM = os.environ.get("OFASR_CONV_SIDE_STREAM", "0") != "0"
# The static convs' weight gradients on the library's side stream, joined with the composite blocks' at the end of the
# backward pass (needs DEFER_WGRAD).  On for the large-plane decoder convs (H*W >= OFASR_CONV_DEFER_MIN_HW): in round 1
# this measured 2213 against 2262 images/s -- the side stream was then full of the VALU-bound depthwise weight gradient --
# and is worth +1.1 % now that it has slack (2439-2445 against 2410-2419, three A/B pairs; all convs deferred: the same).
# OFASR_CONV_DEFER_WGRAD=0 keeps them on the caller's stream.
CONV_DEFER_WGRAD = os.environ.get("OFASR_CONV_DEFER_WGRAD", "1") != "0"
CONV_DEFER_MIN_HW = int(os.environ.get("OFASR_CONV_DEFER_MIN_HW", "10000"))
# deferred conv weight gradients held back until the MB stack's backward starts (beside bandwidth-bound kernels instead of
# the tail's own matrix-bound ones).  Off: 2630-2663 against 2668-2682 images/s (two A/B pairs) -- the tail's BatchNorm
# passes get faster, the MB chain slower by more
CONV_WGRAD_LATE = os.environ.get("OFASR_CONV_WGRAD_LATE", "0") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


CONV_FORCE_HIP = False   # tests: every qualifying conv through the HIP kernel regardless of the measured policy
HIP_CONV = True   # static ConvLayer convolutions (16-bit activations) through the implicit-GEMM HIP kernel


def _conv_hip_ok(x, weight, stride, padding, dilation, groups):
    k = weight.shape[-1]
    return (HIP_CONV and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16, torch.float32)
            and weight.dtype == torch.float32
            and weight.shape[2] == weight.shape[3] and k in (3, 5) and stride == (1, 1) and dilation == (1, 1)
            and groups == 1 and padding == (k // 2, k // 2))


CONV_F32_VENDOR = os.environ.get("OFASR_CONV_F32_VENDOR", "0") != "0"


def _conv_f32_policy(cin, cout, k, H, W):
    """(forward, input gradient, weight gradient) on the own fp32 kernels.  Default: all three, always.

    Measured on MI355X at the S4 shapes (tools/kbench.py --dtype f32, profiles/r02_kbench_f32.txt): the wide convs'
    forward / input gradient run at the vendor kernels' rate (120 against 122 TFLOP/s on 64 -> 256 @128x128, 76 % of
    the fp32 matrix peak); the wide convs' weight gradient (67 against 119) is slower.  The 3-channel stem / head convs
    run on the thin-side kernels of csrc/conv_thin.hip (round 3: head forward 1752 -> 195 us, weight gradient 1656 ->
    136 us; the vendor's were ~1100 us each).  OFASR_CONV_F32_VENDOR=1 selects the vendor's weight gradient for the wide
    convs of regular training shapes; it is not the default because the vendor library's per-shape kernel search costs minutes on a fresh
    machine (260 s for the five shapes of one bench.py run) and because ragged Set14 sizes would search per image."""
    if not CONV_F32_VENDOR or CONV_FORCE_HIP or W % 8 != 0 or H % 2 != 0:
        return True, True, True
    if min(cin, cout) <= 4:     # the 3-channel stem / head: csrc/conv_thin.hip (10x the vendor kernels' rate, round 3)
        return True, True, True
    return True, True, False


class Conv2dF32Fn(Function):
    """nn.Conv2d of the static ConvLayer with fp32 activations: forward, input and weight gradients on the fp32 matrix
    instruction (csrc/conv2d_f32.hip) -- the reference's arithmetic, any H / W -- or, per _conv_f32_policy, the vendor
    convolution where that is measured faster."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        weight = weight.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        ctx.policy = _conv_f32_policy(Cin, Cout, K, H, W)
        ctx.save_for_backward(x, weight)
        if not ctx.policy[0]:
            return torch.nn.functional.conv2d(x, weight, padding=K // 2)
        L = _C.lib()
        y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=x.device)
        wst, wsp, wsn = _ws(L.ofasr_conv2d_f32_workspace(Cin, Cout, K, 0), x.device)
        with _timed("conv2d_f32_fwd_%dto%d_k%d" % (Cin, Cout, K), (x.numel() + y.numel()) * 4, 2 * N * H * W * Cin * Cout * K * K):
            _C.check(L.ofasr_conv2d_f32_fwd(_p(x), _p(weight), _p(y), N, Cin, Cout, H, W, K, wsp, wsn, _stream()),
                     "conv2d_f32_fwd")
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        dy = dy.contiguous()
        L = _C.lib()
        dx = dw = None
        _, dgrad_own, wgrad_own = ctx.policy
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if (need_dx and not dgrad_own) or (need_dw and not wgrad_own):
            vdx, vdw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [K // 2, K // 2], [1, 1], False,
                                                              [0, 0], 1, [need_dx and not dgrad_own,
                                                                          need_dw and not wgrad_own, False])
            if need_dx and not dgrad_own:
                dx, need_dx = vdx, False
            if need_dw and not wgrad_own:
                dw, need_dw = vdw, False
        defer = False
        if need_dw:
            dw = torch.empty_like(weight)
            wst2, wsp2, wsn2 = _ws(L.ofasr_conv2d_f32_wgrad_workspace(N, Cin, Cout, H, W, K), x.device)
            # as Conv2dFn: the weight gradient of a large-plane conv goes to the library's side stream and is joined with the
            # composite blocks' at the end of the backward pass
            side = None
            if CONV_DEFER_WGRAD and H * W >= CONV_DEFER_MIN_HW and _lib_side_stream(x.device) is not None:
                defer = _defer_this_backward((weight,))
                if defer:
                    side = _lib_side_stream(x.device)
                    side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream(x.device)):
                _C.check(L.ofasr_conv2d_f32_wgrad(_p(dy), _p(x), _p(dw), N, Cin, Cout, H, W, K, wsp2, wsn2, _stream()),
                         "conv2d_f32_wgrad")
        if need_dx:
            dx = torch.empty_like(x)
            wst, wsp, wsn = _ws(L.ofasr_conv2d_f32_workspace(Cin, Cout, K, 1), x.device)
            _C.check(L.ofasr_conv2d_f32_dgrad(_p(dy), _p(weight), _p(dx), N, Cin, Cout, H, W, K, wsp, wsn, _stream()),
                     "conv2d_f32_dgrad")
        if defer:
            _Deferred.keep.append((x, dy, dw, wst2, weight))
            _Deferred.grads.append((weight, dw))
            _Deferred.ext_used.add(torch.device(x.device).index or 0)
            return dx, None
        return dx, dw


class Conv2dFn(Function):
    """nn.Conv2d of the static ConvLayer (reference ofa/layers.py:131-151) for 16-bit activations: forward, input
    and weight gradients on the implicit-GEMM MFMA kernels (csrc/conv2d.hip)."""

    @staticmethod
    def forward(ctx, x, weight, dgrad_hip=True):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        ctx.dgrad_hip = dgrad_hip
        L = _C.lib()
        y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device)
        wst, wsp, wsn = _ws(L.ofasr_conv2d_workspace(Cin, Cout, K, 0), x.device)
        with _timed("conv2d_fwd_%dto%d_k%d" % (Cin, Cout, K), (x.numel() + y.numel()) * x.element_size(),
                    2 * N * H * W * Cin * Cout * K * K):
            _C.check(L.ofasr_conv2d_fwd(_p(x), _p(weight), _p(y), N, Cin, Cout, H, W, K, _dt(x), wsp, wsn, _stream()),
                     "conv2d_fwd")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _conv2d_backward(ctx, dy)


class Conv2dStatFn(Function):
    """Conv2dFn whose forward also returns the BatchNorm statistics partials of its output (ofasr_conv2d_fwd_stat:
    [Cout][units] (sum, sum of squares) from the kernel's epilogue) for BNActCPFn -- no pass over the conv output for
    statistics."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        ctx.dgrad_hip = True
        L = _C.lib()
        y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device)
        units = L.ofasr_conv2d_stat_units(N, Cin, Cout, H, W, K)
        partial = torch.empty((Cout, units, 2), dtype=torch.float32, device=x.device)
        wst, wsp, wsn = _ws(L.ofasr_conv2d_workspace(Cin, Cout, K, 0), x.device)
        with _timed("conv2d_fwd_%dto%d_k%d" % (Cin, Cout, K), (x.numel() + y.numel()) * x.element_size(),
                    2 * N * H * W * Cin * Cout * K * K):
            _C.check(L.ofasr_conv2d_fwd_stat(_p(x), _p(weight), _p(y), N, Cin, Cout, H, W, K, _dt(x), _p(partial), units,
                                             wsp, wsn, _stream()), "conv2d_fwd_stat")
        ctx.save_for_backward(x, weight)
        ctx.mark_non_differentiable(partial)
        return y, partial

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _dpartial):
        return _conv2d_backward(ctx, dy)[:2]


def _conv2d_backward(ctx, dy):
    """input / weight gradient of the static conv (shared by Conv2dFn and Conv2dStatFn)"""
    x, weight = ctx.saved_tensors
    N, Cin, H, W = x.shape
    Cout, _, K, _ = weight.shape
    dy = dy.contiguous()
    L = _C.lib()
    dx = dw = None
    if not ctx.dgrad_hip:
        w16 = weight.to(x.dtype)
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w16, None, [1, 1], [K // 2, K // 2], [1, 1], False,
                                                        [0, 0], 1, [ctx.needs_input_grad[0],
                                                                    ctx.needs_input_grad[1], False])
        return dx, (dw.float() if dw is not None else None), None
    # weight gradient on a side stream beside the input gradient (both read dy; neither alone keeps HBM busy
    # through its load / drain phases).  Buffers are allocated on the current stream; fork before, join after.
    cur = torch.cuda.current_stream(x.device)
    side = _side_stream(x.device) if (ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and SIDE_STREAM) else None
    # deferred mode: the weight gradient goes to the library's side stream and is joined (with the composite
    # blocks' weight gradients) once at the end of the backward pass -- it overlaps whatever backward runs next
    defer = False
    if ctx.needs_input_grad[1] and side is None and CONV_DEFER_WGRAD and H * W >= CONV_DEFER_MIN_HW \
            and _lib_side_stream(x.device) is not None:
        defer = _defer_this_backward((weight,))
        if defer:
            side = _lib_side_stream(x.device)
    if ctx.needs_input_grad[1]:
        dw = torch.empty_like(weight)
        wst2, wsp2, wsn2 = _ws(L.ofasr_conv2d_wgrad_workspace(N, Cin, Cout, H, W, K), x.device)
        dt_code = _dt(x)

        def launch_wgrad():
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side if side is not None else cur):
                with _timed("conv2d_wgrad_%dto%d_k%d" % (Cin, Cout, K), (x.numel() + dy.numel()) * x.element_size(),
                            2 * N * H * W * Cin * Cout * K * K):
                    _C.check(L.ofasr_conv2d_wgrad(_p(dy), _p(x), _p(dw), N, Cin, Cout, H, W, K, dt_code, wsp2, wsn2,
                                                  _stream()), "conv2d_wgrad")

        if defer and CONV_WGRAD_LATE:
            # (opt-in, measured slower) not now: launched when the backward pass reaches the MB stack
            # (FusedMBStackFn.backward) or at the flush at the latest
            _Deferred.late.append(launch_wgrad)
        else:
            launch_wgrad()
    if ctx.needs_input_grad[0]:
        dx = torch.empty_like(x)
        wst, wsp, wsn = _ws(L.ofasr_conv2d_workspace(Cin, Cout, K, 1), x.device)
        with _timed("conv2d_dgrad_%dto%d_k%d" % (Cout, Cin, K), (x.numel() + dy.numel()) * x.element_size(),
                    2 * N * H * W * Cin * Cout * K * K):
            _C.check(L.ofasr_conv2d_dgrad(_p(dy), _p(weight), _p(dx), N, Cin, Cout, H, W, K, _dt(x), wsp, wsn,
                                          _stream()), "conv2d_dgrad")
    if defer:
        _Deferred.keep.append((x, dy, dw, wst2, weight))
        _Deferred.grads.append((weight, dw))
        _Deferred.ext_used.add(torch.device(x.device).index or 0)
        return dx, None, None
    if side is not None:
        cur.wait_stream(side)
    return dx, dw, None


ACT_PIXEL_SHUFFLE2 = 2   # ofasr_conv2d_infer_run's act codes: 0 none, 1 ReLU6 (= ACT_RELU6), 2 PixelShuffle(2) store
CONV_BN_EPILOGUE = os.environ.get("OFASR_CONV_BN_EPILOGUE", "1") != "0"
PS_BWD_FUSED = os.environ.get("OFASR_PS_BWD_FUSED", "1") != "0"   # BN backward reads dout through the inverse PixelShuffle


class batched_counters(object):
    """inside this context the `num_batches_tracked += 1` of the static ConvLayers' BatchNorms (one tiny launch each on
    the critical stream: 6 per S4 forward) are collected and applied by ONE multi-tensor add when the context exits
    (OFAMobileNetS4.forward wraps itself in it; a ConvLayer called on its own keeps the immediate bump)"""
    pending = None

    def __enter__(self):
        self.outer = batched_counters.pending
        batched_counters.pending = []
        return self

    def __exit__(self, *exc):
        lst, batched_counters.pending = batched_counters.pending, self.outer
        if lst:
            torch._foreach_add_(lst, 1)
        return False


def conv_bn_act_train(x, conv, bn, act):
    """training-mode ConvLayer: conv (BatchNorm statistics from its epilogue) -> BatchNorm apply (+ ReLU6, or with the
    PixelShuffle(2) as its store), act in {ACT_NONE, ACT_RELU6, ACT_PIXEL_SHUFFLE2}.  One pass less over the conv output
    per layer (two with the shuffle) than conv -> statistics -> apply -> shuffle.  Returns None when the layer is outside
    what the kernels implement (16-bit activations, W % 8 == 0, BN tracking running statistics): the caller composes
    conv2d + bn_act + activation."""
    w = conv.weight
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype("cuda"))
    if (not CONV_BN_EPILOGUE or not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16) or conv.bias is not None
            or w.dtype != torch.float32 or not _conv_hip_ok(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            or x.shape[3] % 8 or not bn.training or not bn.track_running_stats or bn.weight is None
            or bn.num_features != w.shape[0] or (act == ACT_PIXEL_SHUFFLE2 and w.shape[0] % 4)):
        return None
    factor = 0.0
    if bn.num_batches_tracked is not None:
        if batched_counters.pending is not None and bn.momentum is not None:
            batched_counters.pending.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    y, partial = Conv2dStatFn.apply(x, w)
    return BNActCPFn.apply(y, partial, bn.weight, bn.bias, bn.running_mean, bn.running_var, factor, bn.eps, act)


def conv_bn_act_infer(x, conv, bn, act):
    """eval-mode ConvLayer as ONE kernel (ofasr_conv2d_infer_prepare / _run): act(BN_eval(conv(x))), act in
    {ACT_NONE, ACT_RELU6, ACT_PIXEL_SHUFFLE2}; bn may be None.  Returns None when the conv is outside what the kernel
    implements (the caller then composes conv2d + bn_act + activation)."""
    w = conv.weight
    if torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype("cuda"))
    if (not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16) or conv.bias is not None or w.dtype != torch.float32
            or not _conv_hip_ok(x, w, conv.stride, conv.padding, conv.dilation, conv.groups)
            or (bn is not None and (bn.training or bn.weight is None or not bn.track_running_stats))):
        return None
    N, Cin, H, W = x.shape
    Cout, _, K, _ = w.shape
    if act == ACT_PIXEL_SHUFFLE2 and Cout % 4:
        return None
    L = _C.lib()
    xa = x.contiguous()
    padw = (-W) % 8          # ragged widths: zero columns on the right ARE the convolution's padding (see conv2d)
    if padw:
        xa = torch.nn.functional.pad(xa, (0, padw))
    Wp = W + padw
    tensors = (w,) + ((bn.weight, bn.bias, bn.running_mean, bn.running_var) if bn is not None else ())

    def prepare(ptr, nbytes):
        g = [_p(t) for t in tensors[1:]] if bn is not None else [None] * 4
        _C.check(L.ofasr_conv2d_infer_prepare(_p(w), g[0], g[1], g[2], g[3], float(bn.eps) if bn is not None else 0.0, Cin,
                                              Cout, K, _dt(xa), ptr, nbytes, _stream()), "conv2d_infer_prepare")

    kind = ("conv", xa.dtype, Cin, Cout, K, float(bn.eps) if bn is not None else None)
    opnd = _infer_operands(kind, tensors, L.ofasr_conv2d_infer_operand_bytes(Cin, Cout, K), xa.device, prepare)
    if act == ACT_PIXEL_SHUFFLE2:
        y = torch.empty((N, Cout // 4, 2 * H, 2 * Wp), dtype=xa.dtype, device=xa.device)
    else:
        y = torch.empty((N, Cout, H, Wp), dtype=xa.dtype, device=xa.device)
    with _timed("conv2d_infer_%dto%d_k%d" % (Cin, Cout, K), (xa.numel() + y.numel()) * xa.element_size(),
                2 * N * H * Wp * Cin * Cout * K * K):
        _C.check(L.ofasr_conv2d_infer_run(_p(xa), _p(y), N, Cin, Cout, H, Wp, K, _dt(xa), int(act), _p(opnd),
                                          ctypes.c_size_t(opnd.numel()), _stream()), "conv2d_infer_run")
    if padw:
        y = y[..., :(2 * W if act == ACT_PIXEL_SHUFFLE2 else W)]
    return y


def _conv_policy(cin, cout, k):
    """(forward on HIP, backward-data on HIP) -- measured on MI355X (tools/kbench.py, profiles/r01_kbench.txt): the
    implicit-GEMM kernel beats the vendor kernels (which also pay NCHW<->NHWC transposes) on every static conv."""
    return True, True


def conv2d(x, conv):
    """forward of the nn.Conv2d module `conv`: HIP implicit GEMM when the shape/dtype qualifies, else the module."""
    if conv.bias is None and torch.is_autocast_enabled() and x.is_cuda and x.dtype == torch.float32:
        xa = x.to(torch.get_autocast_dtype("cuda"))
    else:
        xa = x
    if conv.bias is None and _conv_hip_ok(xa, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups):
        w = conv.weight
        fwd_hip, bwd_hip = _conv_policy(w.shape[1], w.shape[0], w.shape[2])
        if xa.dtype == torch.float32:       # the reference's arithmetic: exact fp32 on the fp32 matrix instruction
            return Conv2dF32Fn.apply(xa, w)
        if fwd_hip or CONV_FORCE_HIP:
            # the kernel moves rows in 16-byte pieces (W % 8 == 0).  Ragged widths (Set14: 125, 62, 146 ...) are
            # zero-padded on the right to the next multiple of 8 and the extra output columns dropped: the pad columns
            # ARE the convolution's zero padding, so the kept columns are unchanged
            W = xa.shape[3]
            padw = (-W) % 8
            if padw:
                xa = torch.nn.functional.pad(xa, (0, padw))
            y = Conv2dFn.apply(xa, w, bwd_hip or CONV_FORCE_HIP)
            return y[..., :W] if padw else y
    return conv(x)
