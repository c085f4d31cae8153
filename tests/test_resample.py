"""The oracle of the GPU bicubic resize (oracle/pil_bicubic.py: PIL's 8-bit fixed-point resampling restated) against
(a) the REFERENCE's own LR images (tests/golden/div2k.npz: ModCrop(4) + Scale(1/2), Scale(1/4) through the reference's
div2k_setxx.py on PIL) and (b) the Pillow installed here, bit for bit, on sizes that exercise the clipped borders.
CPU-only."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def test_oracle_matches_reference_lr_images(golden):
    from oracle import pil_bicubic
    g = golden("div2k.npz")
    H = np.ascontiguousarray(g["H"].transpose(2, 0, 1))           # HWC uint8 -> planar
    for key, f in (("L2", 2), ("L4", 4)):
        got = pil_bicubic.scale_down(H, f)
        assert np.array_equal(got, g[key].transpose(2, 0, 1)), key


@pytest.mark.parametrize("hw", [(36, 48), (32, 32), (40, 24), (17, 23), (8, 8), (96, 100)])
@pytest.mark.parametrize("factor", [2, 4])
def test_oracle_matches_pillow(hw, factor):
    from oracle import pil_bicubic
    rng = np.random.RandomState(hw[0] * 131 + hw[1] + factor)
    a = rng.randint(0, 256, size=hw + (3,)).astype(np.uint8)
    oh, ow = int(hw[0] * (1.0 / factor)), int(hw[1] * (1.0 / factor))
    ref = np.asarray(Image.fromarray(a, "RGB").resize((ow, oh), Image.BICUBIC))
    got = pil_bicubic.resize_u8(a.transpose(2, 0, 1), oh, ow).transpose(1, 2, 0)
    assert np.array_equal(got, ref)


def test_coefficient_tables_are_periodic_for_integer_factors():
    """what the GPU kernel relies on: for in = out * f the interior taps are the same for every output index"""
    from oracle import pil_bicubic
    for f in (2, 4):
        xmin, cnt, kk = pil_bicubic.coeffs(64 * f, 64)
        inner = slice(4, 60)
        assert np.all(cnt[inner] == cnt[10]) and np.all(kk[inner] == kk[10])
        assert np.all(np.diff(xmin[inner]) == f)
        assert int(kk[10].sum()) in range((1 << 22) - 8, (1 << 22) + 9)
