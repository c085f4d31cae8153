"""GPU parity of the PIL-exact bicubic resize (ofasr_bicubic_resize_u8, csrc/resample.hip) against the oracle
(oracle/pil_bicubic.py, itself pinned to Pillow and to the reference's LR images by tests/test_resample.py) and against
the reference's own outputs (tests/golden/div2k.npz): BIT-EXACT uint8.  Also the provider's uint8-only mode end to end:
the batch the trainer sees equals the batch the host-side PIL path builds (div2k_setxx.py:288-298).  `-m gpu`."""
import os

import numpy as np
import pytest
import torch

from conftest import amd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_bicubic_matches_reference_lr_images(golden):
    ops = amd("ops")
    g = golden("div2k.npz")
    Hi = torch.from_numpy(np.ascontiguousarray(g["H"].transpose(2, 0, 1))).to(DEV)
    h, w = Hi.shape[-2:]
    for key, f in (("L2", 2), ("L4", 4)):
        got = ops.bicubic_resize_u8(Hi, int(h * (1.0 / f)), int(w * (1.0 / f))).cpu().numpy()
        assert np.array_equal(got, g[key].transpose(2, 0, 1)), key


@pytest.mark.parametrize("shape", [(16, 3, 256, 256), (2, 3, 36, 48), (1, 3, 17, 23), (3, 3, 500, 480), (1, 1, 8, 8)])
@pytest.mark.parametrize("factor", [2, 4])
def test_bicubic_vs_oracle_bit_exact(shape, factor):
    from oracle import pil_bicubic
    ops = amd("ops")
    rng = np.random.RandomState(shape[2] * 7 + shape[3] + factor)
    a = rng.randint(0, 256, size=shape).astype(np.uint8)
    oh, ow = int(shape[2] * (1.0 / factor)), int(shape[3] * (1.0 / factor))
    if oh == 0 or ow == 0:
        pytest.skip("empty output")
    got = ops.bicubic_resize_u8(torch.from_numpy(a).to(DEV), oh, ow).cpu().numpy()
    ref = pil_bicubic.resize_u8(a, oh, ow)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_non_integer_and_upscale_factors_vs_oracle():
    from oracle import pil_bicubic
    ops = amd("ops")
    a = np.random.RandomState(3).randint(0, 256, size=(2, 3, 37, 50)).astype(np.uint8)
    for oh, ow in ((18, 25), (9, 12), (37, 20), (40, 50), (74, 100)):
        got = ops.bicubic_resize_u8(torch.from_numpy(a).to(DEV), oh, ow).cpu().numpy()
        assert np.array_equal(got, pil_bicubic.resize_u8(a, oh, ow)), (oh, ow)


def same_u8(dev_t, host_t, what):
    """the uint8 images behind the two float tensors are identical; the floats agree to the last bit or one ulp (the
    /255 of ToTensor runs on the GPU in one path and on the host in the other)"""
    a, b = dev_t.cpu(), host_t
    assert a.shape == b.shape, what
    assert torch.equal((a * 255.0).round().to(torch.uint8), (b * 255.0).round().to(torch.uint8)), what
    assert float((a - b).abs().max()) <= 6e-8, what


def test_provider_uint8_mode_gives_the_same_batches(tmp_path):
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    dp = amd("imagenet_codebase.data_providers.div2k_setxx")
    utils = amd("utils")
    rng = np.random.RandomState(0)
    for split, n in (("train", 4), ("val", 3)):
        os.makedirs(os.path.join(str(tmp_path), split))
        for i in range(n):
            a = rng.randint(0, 256, size=(40 + 4 * i, 52 + 4 * i, 3)).astype(np.uint8)
            Image.fromarray(a, "RGB").save(os.path.join(str(tmp_path), split, "%02d.png" % i))
    kw = dict(save_path=str(tmp_path), train_batch_size=2, test_batch_size=1, n_worker=0, image_size=32)
    host = dp.Div2K_SetXXDataProvider(**kw)
    dev = dp.Div2K_SetXXDataProvider(lr_on_device=True, **kw)
    for hb, db in zip(host.test, dev.test):          # full ModCrop(4) images, deterministic
        assert list(db.keys()) == ["image_u8"]
        got = utils.device_batch(db, DEV)
        for k in ("image", "2x_down_image", "4x_down_image"):
            same_u8(got[k], hb[k], k)
    torch.manual_seed(5)
    hb = next(iter(host.train))
    torch.manual_seed(5)
    got = utils.device_batch(next(iter(dev.train)), DEV)   # same RNG draws => same random crops / flips / rotations
    for k in ("image", "2x_down_image", "4x_down_image"):
        same_u8(got[k], hb[k], k)
