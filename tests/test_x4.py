"""OFAMobileNetX4 (learned down-scaler + SR autoencoder): structure / sampling parity on CPU and
forward+backward parity with the reference golden on the GPU."""
import json
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, amd, assert_close
from detfill import det_uniform, fill_state_dict


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(GOLDEN, "x4_meta.json")))


def _build():
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = amd("elastic_nn.networks").OFAMobileNetX4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6],
                                                    depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    return net   # NB the flag is class-level and also read at forward time (reference dynamic_op.py:52): leave it set


def test_x4_structure_and_sampling(meta):
    net = _build()
    assert {k: list(v.shape) for k, v in net.state_dict().items()} == meta["state_dict_shapes"]
    assert [n for n, _ in net.named_parameters()] == meta["param_names"]
    assert net.block_group_info == meta["block_group_info"]
    assert sum(p.numel() for p in net.parameters()) == meta["n_params"]
    for t in meta["sample_traces"]:
        random.seed(t["seed"])
        assert net.sample_active_subnet() == t["sampled"]
        assert net.runtime_depth == t["runtime_depth"]
    for st, rd in zip(meta["settings"], meta["runtime_depth"]):
        net.set_active_subnet(**st)
        assert net.runtime_depth == rd
    # compat: encoder stage 0 and decoder stage 0 both read runtime_depth[0] (= pixel_d)
    net.set_active_subnet(ks=3, e=3, d=4, pixel_d=1)
    kinds = [k for k, _ in net.active_block_sequence()]
    assert kinds.count("unshuffle") == 1 and kinds.count("shuffle") == 1 and kinds.count("mb") == 2 * (1 + 4 + 4 + 4)


@pytest.mark.gpu
@pytest.mark.parametrize("si", [0, 1, 2])
def test_x4_forward_backward_golden(golden, meta, si):
    g = golden("x4_net.npz")
    net = _build()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(shapes, "x4").items()})
    net.to("cuda:0").train()
    net.set_active_subnet(**meta["settings"][si])
    x = torch.from_numpy(det_uniform((2, 3, 24, 16), "x4/hr", 0.0, 1.0)).to("cuda:0")
    y = net(x)
    assert tuple(y.shape) == tuple(x.shape)          # autoencoder: output size == input size (Q3)
    assert_close(y.detach().cpu().numpy(), g["y_s%d" % si], 5e-4, 5e-5, "y")
    loss = F.mse_loss(y, x)
    assert abs(float(loss.detach()) - float(g["loss_s%d" % si])) <= 2e-5 * abs(float(g["loss_s%d" % si]))
    loss.backward()
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    params = dict(net.named_parameters())
    names = meta["param_names"]
    assert np.array_equal(np.array([params[n].grad is None for n in names]), g["g_isnone_s%d" % si])
    l2 = np.array([0.0 if params[n].grad is None else float(params[n].grad.double().pow(2).sum().sqrt())
                   for n in names])
    assert_close(l2, g["g_l2_s%d" % si], 5e-3, 1e-7, "grad l2 norms")
