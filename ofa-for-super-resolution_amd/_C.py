"""ctypes binding of libofasr_hip.so (C ABI: include/ofasr.h).

This is the whole FFI layer: plain pointers and sizes in, status code out.  The library is
built in-tree by csrc/build.sh (hipcc --offload-arch=gfx950); it is NOT built lazily at import
on a GPU box -- `__graft_entry__.build()` does it -- and a missing library is a hard error:
there is no fallback path.
"""
import ctypes
import os
import subprocess

# torch first: it ships its own libamdhip64; loading it before libofasr_hip.so makes both share ONE HIP
# runtime (same device context, same streams).  The other order leaves two runtimes in the process and
# every launch from this library fails with "no ROCm-capable device is detected".
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFASR_LIB_PATH") or os.path.join(_HERE, "csrc", "libofasr_hip.so")   # override: A/B of two builds on one box

F32, F16, BF16 = 0, 1, 2

_lib = None

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_vp = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); every symbol declared in include/ofasr.h
SIGNATURES = {
    "ofasr_version": (_c_int, []),
    "ofasr_last_error_string": (ctypes.c_char_p, []),
    "ofasr_status_string": (ctypes.c_char_p, [_c_int]),
    "ofasr_pixel_shuffle": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp]),
    "ofasr_pixel_unshuffle": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp]),
    "ofasr_ktransform_fwd": (_c_int, [_c_vp, ctypes.POINTER(_c_int), _c_int, ctypes.POINTER(_c_vp), _c_int,
                                      _c_vp, _c_i64, _c_vp]),
    "ofasr_ktransform_bwd_workspace": (_c_sz, [ctypes.POINTER(_c_int), _c_int, _c_i64]),
    "ofasr_ktransform_bwd": (_c_int, [_c_vp, ctypes.POINTER(_c_int), _c_int, ctypes.POINTER(_c_vp), _c_int, _c_vp,
                                      _c_vp, ctypes.POINTER(_c_vp), _c_i64, _c_vp, _c_sz, _c_vp]),
    "ofasr_dwconv_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp]),
    "ofasr_dwconv_dgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp]),
    "ofasr_dwconv_wgrad_workspace": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_int]),
    "ofasr_dwconv_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp,
                                    _c_sz, _c_vp]),
    "ofasr_pwconv_fwd": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp]),
    "ofasr_pwconv_dgrad": (_c_int, [_c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp]),
    "ofasr_pwconv_wgrad_workspace": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64]),
    "ofasr_pwconv_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp,
                                    _c_sz, _c_vp]),
    "ofasr_bn_workspace": (_c_sz, [_c_i64, _c_i64]),
    "ofasr_bn_partials": (_c_int, [_c_i64, _c_i64]),
    "ofasr_bn_stats": (_c_int, [_c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_bn_finalize": (_c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_vp,
                                   ctypes.c_double, ctypes.c_double, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "ofasr_bn_act_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int,
                                  _c_vp]),
    "ofasr_bn_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_double, ctypes.c_double, _c_int,
                              _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_bn_act_bwd_workspace": (_c_sz, [_c_i64, _c_i64]),
    "ofasr_bn_act_bwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                  _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_conv2d_workspace": (_c_sz, [_c_i64, _c_i64, _c_int, _c_int]),
    "ofasr_conv2d_stat_units": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int]),
    "ofasr_conv2d_fwd_stat": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp,
                                       _c_i64, _c_vp, _c_sz, _c_vp]),
    "ofasr_bn_fwd_cp": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_double,
                                 ctypes.c_double, _c_int, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp]),
    "ofasr_bn_finalize_cp": (_c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_double,
                                      ctypes.c_double, _c_int, _c_vp, _c_vp]),
    "ofasr_pixel_shuffle2_bn": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp]),
    "ofasr_bn_bwd_ps2_workspace": (_c_sz, [_c_i64, _c_i64]),
    "ofasr_bn_bwd_ps2": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_int,
                                  _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_conv2d_infer_operand_bytes": (_c_sz, [_c_i64, _c_i64, _c_int]),
    "ofasr_conv2d_infer_prepare": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_double, _c_i64, _c_i64, _c_int, _c_int,
                                            _c_vp, _c_sz, _c_vp]),
    "ofasr_conv2d_infer_run": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_vp,
                                        _c_sz, _c_vp]),
    "ofasr_conv2d_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp,
                                  _c_sz, _c_vp]),
    "ofasr_conv2d_dgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp,
                                    _c_sz, _c_vp]),
    "ofasr_conv2d_wgrad_workspace": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int]),
    "ofasr_conv2d_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_vp,
                                    _c_sz, _c_vp]),
    "ofasr_conv2d_f32_workspace": (_c_sz, [_c_i64, _c_i64, _c_int, _c_int]),
    "ofasr_conv2d_f32_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_conv2d_f32_dgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_conv2d_f32_wgrad_workspace": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int]),
    "ofasr_conv2d_f32_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbconv_workspace": (_c_sz, [_c_vp]),
    "ofasr_mbconv_act_elems": (_c_sz, [_c_vp]),
    "ofasr_mbconv_stat_floats": (_c_sz, [_c_vp]),
    "ofasr_mbconv_fwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbconv_bwd": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbstack_fwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp]),
    "ofasr_mbstack_bwd": (_c_int, [_c_vp, _c_int, _c_vp, _c_vp, _c_vp]),
    "ofasr_mbconv_defer_join": (_c_int, [_c_int]),
    "ofasr_mbconv_join": (_c_int, [_c_vp]),
    "ofasr_side_stream": (_c_vp, []),
    "ofasr_bicubic_resize_u8_workspace": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "ofasr_bicubic_resize_u8": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbconv_infer_supported": (_c_int, [_c_vp]),
    "ofasr_mbconv_infer_workspace": (_c_sz, [_c_vp]),
    "ofasr_mbconv_infer": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbconv_infer_operand_bytes": (_c_sz, [_c_vp]),
    "ofasr_mbconv_infer_scratch_bytes": (_c_sz, [_c_vp]),
    "ofasr_mbconv_infer_prepare": (_c_int, [_c_vp, _c_vp, _c_sz, _c_vp]),
    "ofasr_mbconv_infer_run": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_sz, _c_vp, _c_sz, _c_vp]),
    "ofasr_debug_mbfused_tile": (_c_int, [_c_int]),
    "ofasr_debug_mbfused_split": (_c_int, [_c_int]),
    "ofasr_debug_mbconv_bn_bwd_stat": (_c_int, [_c_int]),
    "ofasr_debug_launch_count": (ctypes.c_longlong, [ctypes.c_char_p]),
    "ofasr_debug_reset_launch_counts": (None, []),
    "ofasr_debug_launch_table": (ctypes.c_char_p, []),
    "ofasr_profile_enable": (_c_int, [_c_int]),
    "ofasr_profile_read": (ctypes.c_char_p, []),
}


def launch_count(substr=""):
    """launches of the kernels whose symbol contains `substr` since the last reset (include/ofasr.h, Diagnostics)."""
    return int(lib().ofasr_debug_launch_count(substr.encode()))


def reset_launch_counts():
    lib().ofasr_debug_reset_launch_counts()


def launch_table():
    out = {}
    for line in lib().ofasr_debug_launch_table().decode().splitlines():
        n, name = line.split("\t", 1)
        if int(n):
            out[name] = int(n)
    return out


def profile_read():
    """{kernel symbol: {launches, total_us, bytes, flops}} of the launches bracketed since the last read."""
    out = {}
    for line in lib().ofasr_profile_read().decode().splitlines():
        name, n, us, by, fl = line.rsplit("\t", 4)
        out[name] = {"launches": float(n), "total_us": float(us), "bytes": float(by), "flops": float(fl)}
    return out


class MBConvDesc(ctypes.Structure):
    """mirror of ofasr_mbconv_desc (include/ofasr.h) -- field order and types must match the C struct"""
    _fields_ = [
        ("N", _c_i64), ("Cin", _c_i64), ("mid", _c_i64), ("Cout", _c_i64), ("H", _c_i64), ("W", _c_i64),
        ("K", _c_int), ("ks", _c_int * 4), ("chain_len", _c_int), ("transform", _c_int), ("dtype", _c_int),
        ("residual", _c_int), ("bn_training", _c_int * 3), ("bn_momentum", ctypes.c_double * 3),
        ("bn_eps", ctypes.c_double * 3), ("Cmid_max", _c_i64), ("Cout_max", _c_i64), ("ldw1", _c_i64),
        ("ldw2", _c_i64), ("w1", _c_vp), ("w2", _c_vp), ("wdw_max", _c_vp), ("mats", _c_vp * 3),
        ("gamma", _c_vp * 3), ("beta", _c_vp * 3), ("running_mean", _c_vp * 3), ("running_var", _c_vp * 3),
        ("num_batches_tracked", _c_vp * 3),
    ]


class MBConvGrads(ctypes.Structure):
    """mirror of ofasr_mbconv_grads"""
    _fields_ = [("dw1", _c_vp), ("dw2", _c_vp), ("dwdw_max", _c_vp), ("dmats", _c_vp * 3), ("dgamma", _c_vp * 3),
                ("dbeta", _c_vp * 3)]


class MBStackItem(ctypes.Structure):
    """mirror of ofasr_mbstack_item"""
    _fields_ = [("desc", _c_vp), ("act_buf", _c_vp), ("stat_buf", _c_vp), ("workspace", _c_vp), ("workspace_bytes", _c_sz),
                ("tmp_buf", _c_vp), ("grads", _c_vp), ("dx", _c_vp)]


class OfasrError(RuntimeError):
    pass


def build(force=False):
    """compile csrc/*.hip -> csrc/libofasr_hip.so (gfx950)."""
    global _lib
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["bash", os.path.join(_HERE, "csrc", "build.sh")])
    _lib = None
    return LIB_PATH


def lib():
    """the loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfasrError(
                "HIP extension %s is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the OFA-SR hot path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        raise OfasrError("%s failed: %s (%s)" % (
            what, L.ofasr_status_string(status).decode(), L.ofasr_last_error_string().decode()))
