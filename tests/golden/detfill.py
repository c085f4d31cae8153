"""Deterministic, platform-independent tensor fill shared by the golden generator and the tests.

Integer (splitmix64-style) hashing of (tag, index) -> uniform float64 in [lo, hi) -> float32.
No libm, no torch RNG: the same bits on every machine, so fixtures only need to hold OUTPUTS.
"""
import zlib

import numpy as np

_M64 = (1 << 64) - 1


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, tag, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(tag.encode("utf-8")))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + seed * np.uint64(0xD1B54A32D192ED03)
        z = _mix(z)
    u = (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_ints(shape, tag, lo=-8, hi=8):
    """integer-valued float32 (exact in bf16/fp16 too) for bit-exact index-map tests."""
    n = int(np.prod(shape))
    seed = np.uint64(zlib.crc32(tag.encode("utf-8")))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        z = _mix((idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + seed)
    v = (z % np.uint64(hi - lo)).astype(np.int64) + lo
    return v.astype(np.float32).reshape(shape)


def fill_state_dict(shapes, prefix="sd"):
    """shapes: {name: shape}.  Conv / matrix weights ~ U(-a, a) scaled by fan-in; BN weight in
    [0.5, 1.5]; BN bias in [-0.3, 0.3]; running_mean in [-0.2, 0.2]; running_var in [0.5, 1.5];
    num_batches_tracked = 0.  Every term of every formula is exercised (non-identity transform
    matrices, non-trivial BN affine and statistics)."""
    out = {}
    for name, shape in shapes.items():
        tag = "%s/%s" % (prefix, name)
        shape = tuple(int(s) for s in shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = det_uniform(shape, tag, -0.2, 0.2)
        elif name.endswith("running_var"):
            out[name] = det_uniform(shape, tag, 0.5, 1.5)
        elif name.endswith("_matrix"):
            q = shape[0]
            out[name] = (np.eye(q, dtype=np.float32) + det_uniform(shape, tag, -0.15, 0.15)).astype(np.float32)
        elif ".bn." in name or name.endswith("bn.weight") or name.endswith("bn.bias"):
            if name.endswith("weight"):
                out[name] = det_uniform(shape, tag, 0.5, 1.5)
            else:
                out[name] = det_uniform(shape, tag, -0.3, 0.3)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            a = float(np.sqrt(3.0 / max(fan_in, 1)))
            out[name] = det_uniform(shape, tag, -a, a)
    return out
