"""Data path of the SR trainer: the PIL-only Div2K_SetXX provider against the reference's own
PIL code (tests/golden/div2k.npz, make_golden.py gen_div2k) and against the documented torchvision draw order of
the three random train transforms.  CPU-only."""
import os

import numpy as np
import pytest
import torch

from conftest import amd

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


@pytest.fixture(scope="module")
def dp():
    return amd("imagenet_codebase.data_providers.div2k_setxx")


def _tree(root, n_train=6, n_val=3, size=(37, 50)):
    rng = np.random.RandomState(0)
    for split, n in (("train", n_train), ("val", n_val)):
        os.makedirs(os.path.join(root, split))
        for i in range(n):
            a = rng.randint(0, 256, size=(size[0] + i, size[1] + 2 * i, 3)).astype(np.uint8)
            Image.fromarray(a, "RGB").save(os.path.join(root, split, "%02d.png" % i))


def test_modcrop_scale_dataset_match_reference(dp, golden, tmp_path):
    g = golden("div2k.npz")
    img = Image.fromarray(g["hr"], "RGB")
    H = dp.ModCrop(mod=4)(img)
    assert np.array_equal(np.asarray(H), g["H"])
    assert np.array_equal(np.asarray(dp.get_transform_L(opt=2)(H)), g["L2"])
    assert np.array_equal(np.asarray(dp.get_transform_L(opt=4)(H)), g["L4"])
    # the dataset: decode -> transform -> two bicubic down-scales -> [0,1] CHW float tensors under the reference's keys
    img.save(str(tmp_path / "x.png"))
    ds = dp.Div2K_SetXXDataset(str(tmp_path), dp.Compose([dp.ModCrop(mod=4)]))
    s = ds[0]
    assert sorted(s.keys()) == ["2x_down_image", "4x_down_image", "image"]
    for key, ref in (("image", g["H"]), ("2x_down_image", g["L2"]), ("4x_down_image", g["L4"])):
        assert s[key].dtype == torch.float32 and tuple(s[key].shape) == (3,) + ref.shape[:2]
        assert torch.equal(s[key], torch.from_numpy(ref.transpose(2, 0, 1).copy()).float() / 255.0)


def test_recursive_listing_matches_reference(dp, golden, tmp_path):
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "sub", "deeper"))
    for rel in ("a.png", "z.txt", os.path.join("sub", "b.png"), os.path.join("sub", "deeper", "c.jpg")):
        open(os.path.join(d, rel), "wb").close()
    listing = [os.path.relpath(q, d) for q in dp.get_image_paths_recursive(d, [])]
    assert "|".join(listing) == str(golden("div2k.npz")["listing"])


def test_train_transform_draw_order(dp):
    """RandomCrop -> RandomHorizontalFlip -> RandomRotation consume the torch global RNG exactly like torchvision:
    randint(h-th+1), randint(w-tw+1), rand(1), uniform_(-90, 90)."""
    a = np.random.RandomState(1).randint(0, 256, size=(40, 56, 3)).astype(np.uint8)
    img = Image.fromarray(a, "RGB")
    t = dp.Compose([dp.RandomCrop(32), dp.RandomHorizontalFlip(), dp.RandomRotation(degrees=(-90, 90))])
    torch.manual_seed(123)
    out = t(img)
    after = torch.rand(1)
    torch.manual_seed(123)
    i = int(torch.randint(0, 40 - 32 + 1, size=(1,)).item())
    j = int(torch.randint(0, 56 - 32 + 1, size=(1,)).item())
    flip = bool(torch.rand(1) < 0.5)
    angle = float(torch.empty(1).uniform_(-90.0, 90.0).item())
    assert torch.equal(after, torch.rand(1))
    ref = img.crop((j, i, j + 32, i + 32))
    if flip:
        ref = ref.transpose(Image.FLIP_LEFT_RIGHT)
    ref = ref.rotate(angle, Image.NEAREST, False, None)
    assert out.size == (32, 32) and np.array_equal(np.asarray(out), np.asarray(ref))


def test_provider_end_to_end(dp, tmp_path):
    _tree(str(tmp_path))
    prov = dp.Div2K_SetXXDataProvider(save_path=str(tmp_path), train_batch_size=2, test_batch_size=1, valid_size=None,
                                      n_worker=0, image_size=32)
    assert prov.name() == "div2k_setxx" and prov.data_shape == (3, 32, 32) and prov.valid is prov.test
    torch.manual_seed(0)
    batches = list(prov.train)
    assert len(batches) == 3   # 6 images, batch 2, drop_last
    b = batches[0]
    assert tuple(b["image"].shape) == (2, 3, 32, 32) and tuple(b["2x_down_image"].shape) == (2, 3, 16, 16)
    assert tuple(b["4x_down_image"].shape) == (2, 3, 8, 8)
    assert 0.0 <= float(b["4x_down_image"].min()) and float(b["image"].max()) <= 1.0
    for vb in prov.test:   # full ModCrop(4) images, batch 1
        _, _, h, w = vb["image"].shape
        assert h % 4 == 0 and w % 4 == 0 and tuple(vb["4x_down_image"].shape[2:]) == (h // 4, w // 4)
    sub = prov.build_sub_train_loader(4, 2, num_worker=0)
    assert len(sub) == 2 and sub is prov.build_sub_train_loader(4, 2, num_worker=0)   # cached
    # validation split + rank partition (MyDistributedSampler over index subsets)
    tr, va = dp.DataProvider.random_sample_valid_set(6, 2)
    assert sorted(tr + va) == list(range(6)) and len(va) == 2
    ds = prov.train.dataset
    parts = []
    for r in range(2):
        smp = dp.MyDistributedSampler(ds, 2, r, np.array(tr))
        smp.set_epoch(3)
        parts.append(list(iter(smp)))
    assert len(parts[0]) == len(parts[1]) == 2 and sorted(parts[0] + parts[1]) == sorted(tr)


def test_run_config_picks_real_provider_when_dataset_exists(dp, tmp_path, monkeypatch):
    rm = amd("imagenet_codebase.run_manager")
    cfg = rm.Div2K_SetXXRunConfig(train_batch_size=2, test_batch_size=1, n_worker=0, image_size=32)
    assert (cfg.distort_color, cfg.image_size) == (None, 32)   # the reference's defaults (run_manager/__init__.py:134)
    monkeypatch.setenv("OFASR_DIV2K_ROOT", str(tmp_path / "missing"))
    monkeypatch.delenv("OFASR_ALLOW_SYNTHETIC_DATA", raising=False)
    with pytest.raises(FileNotFoundError):      # a mistyped dataset path must not train on noise silently
        rm.Div2K_SetXXRunConfig(train_batch_size=2, test_batch_size=1, n_worker=0, image_size=32).data_provider
    cfg0 = rm.Div2K_SetXXRunConfig(train_batch_size=2, test_batch_size=1, n_worker=0, image_size=32,
                                   allow_synthetic=True)
    assert cfg0.data_provider.name() == "synthetic_sr" and "synthetic" in cfg0.dataset and "synthetic" in cfg0.config["dataset"]
    _tree(str(tmp_path))
    monkeypatch.setenv("OFASR_DIV2K_ROOT", str(tmp_path))
    cfg1 = rm.Div2K_SetXXRunConfig(train_batch_size=2, test_batch_size=1, n_worker=0, image_size=32)
    prov = cfg1.data_provider
    assert isinstance(prov, dp.Div2K_SetXXDataProvider) and prov is cfg1.data_provider
    assert tuple(next(iter(cfg1.train_loader))["image"].shape) == (2, 3, 32, 32)


def test_training_script_settings_build_the_real_provider(dp, tmp_path, monkeypatch):
    """the run config exactly as train_ofa_net_sr_simple.py builds it (Div2K_SetXXRunConfig(**args.__dict__) with the
    script's hard-coded settings, reference train_ofa_net_sr_simple.py:88-132) against a tiny on-disk PNG tree: the
    provider constructs and yields a batch (distort_color must be None as in the reference, :115)."""
    import argparse
    import re
    from conftest import ROOT
    rm = amd("imagenet_codebase.run_manager")
    src = open(os.path.join(ROOT, "train_ofa_net_sr_simple.py")).read()
    m = re.search(r"args\.resize_scale, args\.distort_color = ([^\n#]+)", src)
    resize_scale, distort_color = eval(m.group(1))
    assert distort_color is None
    args = argparse.Namespace(
        task="kernel", phase=1, path=str(tmp_path / "exp"), n_epochs=1, base_lr=1e-3, dynamic_batch_size=1,
        manual_seed=0, lr_schedule_type="cosine", base_batch_size=16, valid_size=None, opt_type="adam", momentum=0.9,
        no_nesterov=False, weight_decay=3e-5, label_smoothing=0.0, no_decay_keys="bn#bias", fp16_allreduce=False,
        model_init="he_fout", validation_frequency=1, print_frequency=10, n_worker=0, resize_scale=resize_scale,
        distort_color=distort_color, image_size=32, continuous_size=True, not_sync_distributed_image_size=False,
        bn_momentum=0.1, bn_eps=1e-5, dropout=0.1, width_mult_list="1.0", dy_conv_scaling_mode=1,
        independent_distributed_sampling=False, kd_ratio=0, kd_type="ce", teacher_model=None, warmup_epochs=0,
        warmup_lr=-1, init_lr=1e-3, train_batch_size=2, test_batch_size=1)
    _tree(str(tmp_path / "data"))
    monkeypatch.setenv("OFASR_DIV2K_ROOT", str(tmp_path / "data"))
    cfg = rm.Div2K_SetXXRunConfig(**args.__dict__)
    prov = cfg.data_provider
    assert isinstance(prov, dp.Div2K_SetXXDataProvider)
    b = next(iter(cfg.train_loader))
    assert tuple(b["image"].shape) == (2, 3, 32, 32) and tuple(b["4x_down_image"].shape) == (2, 3, 8, 8)


def test_eval_loaders_are_whole_and_ordered_under_sharding(dp, tmp_path):
    """data parallelism: training indices are split over the ranks and re-shuffled per epoch (set_epoch); validation /
    test images are NOT split -- every rank sees every image once, in file order (no shuffle, no padding duplicates,
    no drop_last), so the logged PSNR does not depend on the rank count."""
    _tree(str(tmp_path), n_train=7, n_val=3)
    provs = [dp.Div2K_SetXXDataProvider(save_path=str(tmp_path), train_batch_size=1, test_batch_size=1, n_worker=0,
                                        image_size=32, num_replicas=2, rank=r) for r in range(2)]
    shapes = [[tuple(b["image"].shape) for b in p.test] for p in provs]
    assert shapes[0] == shapes[1] and len(shapes[0]) == 3
    assert sorted(s[2:] for s in shapes[0]) == [(36, 48), (36, 52), (36, 52)]     # 37x50, 38x52, 39x54 after ModCrop(4)
    assert shapes[0] == [tuple(b["image"].shape) for b in provs[0].test]             # same (listing) order every pass
    per_epoch = []
    for epoch in (0, 1):
        idx = []
        for p in provs:
            p.train.sampler.set_epoch(epoch)
            idx.append(list(iter(p.train.sampler)))
        assert len(idx[0]) == len(idx[1]) == 4 and set(idx[0] + idx[1]) == set(range(7))
        per_epoch.append(idx)
    assert per_epoch[0] != per_epoch[1]
