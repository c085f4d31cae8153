"""Pin the network-level CPU oracle (oracle/s4_port.py) to goldens from the reference
(tests/golden/s4_net.npz, s4_meta.json, mbblock.npz).  CPU only."""
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, assert_close
from detfill import det_uniform, fill_state_dict
from oracle import s4_port


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLDEN, "s4_meta.json")))


def _make_sd(shapes, prefix):
    sd = fill_state_dict(shapes, prefix)
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(v.copy())
        if s4_port.is_param(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def test_state_dict_layout_matches_reference(meta):
    shapes = s4_port.state_dict_shapes()
    ref = {k: tuple(v) for k, v in meta["state_dict_shapes"].items()}
    assert shapes == ref
    assert len(shapes) == 356
    n_params = sum(int(np.prod(s)) for k, s in shapes.items() if s4_port.is_param(k))
    assert n_params == meta["n_params"] == 2160422


def test_sampling_traces(meta):
    arch = s4_port.Arch()
    for t in meta["sample_traces"]:
        random.seed(t["seed"])
        s = arch.sample_active_subnet()
        assert s == t["sampled"]
        assert arch.runtime_depth == t["runtime_depth"]
        assert arch.ks == t["ks"] and arch.e == t["e"]


@pytest.mark.parametrize("si", [0, 1, 2])
@pytest.mark.parametrize("bn_train", [True, False])
def test_s4_forward_backward(golden, meta, si, bn_train):
    g = golden("s4_net.npz")
    shapes = {k: tuple(v) for k, v in meta["state_dict_shapes"].items()}
    sd = _make_sd(shapes, "s4")
    arch = s4_port.Arch()
    arch.set_active_subnet(**meta["settings"][si])
    if bn_train:
        assert arch.runtime_depth == meta["runtime_depth"][si]
    lr = torch.from_numpy(g["lr"])
    y = s4_port.s4_forward(sd, lr, arch, training=bn_train)
    tag = "s%d_%s" % (si, "train" if bn_train else "eval")
    assert_close(y.detach().numpy(), g["y_" + tag], 2e-4, 2e-5, "y")
    hr = torch.from_numpy(det_uniform(tuple(y.shape), "s4/hr%d" % si, 0.0, 1.0))
    loss = F.mse_loss(y, hr)
    assert abs(float(loss) - float(g["loss_" + tag])) <= 1e-5 * abs(float(g["loss_" + tag]))
    loss.backward()
    names = meta["param_names"]
    isnone = np.array([sd[n].grad is None for n in names])
    assert np.array_equal(isnone, g["g_isnone_" + tag])
    l2 = np.array([0.0 if sd[n].grad is None else float(sd[n].grad.double().pow(2).sum().sqrt()) for n in names])
    assert_close(l2, g["g_l2_" + tag], 2e-3, 1e-7, "grad l2 norms")
    for k in g.files:
        if k.startswith("grad_") and k.endswith("_" + tag):
            name = k[len("grad_"):-len("_" + tag)]
            ref = g[k]
            assert_close(sd[name].grad.numpy(), ref, 2e-3, 2e-6 * max(1.0, float(np.abs(ref).max())), name)
    if bn_train:
        for k in g.files:
            if k.startswith("buf_") and k.endswith("_" + tag):
                name = k[len("buf_"):-len("_" + tag)]
                assert_close(sd[name].detach().numpy(), g[k], 1e-5, 1e-6, name)


def test_psnr_through_metric(golden, ora):
    g = golden("s4_net.npz")
    v = ora.psnr_y(g["psnr_y1"], g["psnr_tgt"])
    assert abs(v - float(g["psnr_value"])) < 1e-9
