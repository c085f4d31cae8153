"""CPU restatement of PIL's 8-bit bicubic resize (Pillow src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) -- the arithmetic behind the reference's LR
images: Scale(1/2), Scale(1/4) = img.resize(size, Image.BICUBIC) on uint8 RGB
(/root/reference ofa/imagenet_codebase/data_providers/div2k_setxx.py:354-380, 288-298).

TEST INFRASTRUCTURE ONLY: the checker of the GPU kernel (csrc/resample.hip, ofasr_bicubic_resize_u8).  Pillow is a
third-party dependency of the reference (requirements.txt lists torchvision, which pulls it; un-pinned); this file
restates its published fixed-point algorithm and is pinned (a) by the reference's own outputs in tests/golden/div2k.npz
and (b) against the Pillow installed in the build container (tests/test_resample.py).

Algorithm: separable; horizontal pass first (uint8 intermediate), then vertical.  Per output index xx of an axis
resized from `in_size` to `out_size`:
    scale = in_size / out_size; filterscale = max(scale, 1); support = 2 * filterscale
    center = (xx + 0.5) * scale; xmin = max(int(center - support + 0.5), 0); xmax = min(int(center + support + 0.5), in_size)
    w[x] = bicubic((x + xmin - center + 0.5) / filterscale), normalised to sum 1 in double,
    fixed point: k = int(w * 2**22 +/- 0.5) (round half away from zero, truncating cast)
    out = clip8((sum_x pixel[xmin + x] * k[x] + 2**21) >> 22)
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size, out_size):
    """(xmin[out], count[out], k[out, ksize] int32) exactly as precompute_coeffs + normalize_coeffs_8bpc"""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, np.int32)
    counts = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmins[xx], counts[xx] = xmin, xmax
    return xmins, counts, kk


def _pass(img, out_size, axis):
    """img: uint8 [..., H, W]; resample `axis` (-1 horizontal, -2 vertical) to out_size"""
    img = np.moveaxis(img, axis, -1)
    xmins, counts, kk = coeffs(img.shape[-1], out_size)
    out = np.empty(img.shape[:-1] + (out_size,), np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        n, x0 = int(counts[xx]), int(xmins[xx])
        acc = (src[..., x0:x0 + n] * kk[xx, :n].astype(np.int64)).sum(axis=-1) + (1 << (PRECISION_BITS - 1))
        out[..., xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def resize_u8(img, out_h, out_w):
    """img uint8 [..., H, W] (planar) -> [..., out_h, out_w]; PIL's Image.resize((out_w, out_h), BICUBIC) per plane"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if out_w != img.shape[-1]:
        img = _pass(img, out_w, -1)
    if out_h != img.shape[-2]:
        img = _pass(img, out_h, -2)
    return img


def scale_down(img, factor):
    """the reference's Scale(1/factor): output size (int(h / factor), int(w / factor)) (div2k_setxx.py:359-368)"""
    h, w = img.shape[-2:]
    return resize_u8(img, int(h * (1.0 / factor)), int(w * (1.0 / factor)))
