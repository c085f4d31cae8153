#!/usr/bin/env python3
"""X4 goldens from the reference (operator surface + one autoencoder forward/backward):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_x4.py"""
import json
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

import make_golden as mg   # registers the torchvision stub and puts the reference on sys.path
from detfill import det_uniform, fill_state_dict


def main():
    from ofa.elastic_nn.networks import OFAMobileNetX4
    from ofa.elastic_nn.modules.dynamic_op import DynamicSeparableConv2d
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = OFAMobileNetX4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                         pixelshuffle_depth_list=[1, 2])
    meta = {"state_dict_shapes": {k: list(v.shape) for k, v in net.state_dict().items()},
            "param_names": [n for n, _ in net.named_parameters()],
            "block_group_info": net.block_group_info,
            "n_params": int(sum(p.numel() for p in net.parameters()))}
    traces = []
    for seed in (0, 1000, 7000):
        random.seed(seed)
        s = net.sample_active_subnet()
        traces.append(dict(seed=seed, sampled=s, runtime_depth=list(net.runtime_depth)))
    meta["sample_traces"] = traces
    settings = [dict(ks=7, e=6, d=4, pixel_d=2), dict(ks=3, e=3, d=2, pixel_d=1), dict(ks=5, e=4, d=3, pixel_d=2)]
    meta["settings"] = settings
    meta["runtime_depth"] = []
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = fill_state_dict(shapes, "x4")
    out = {}
    x = det_uniform((2, 3, 24, 16), "x4/hr", 0.0, 1.0)
    for si, setting in enumerate(settings):
        net.load_state_dict({k: mg.T(v) for k, v in sd.items()})
        net.train()
        net.set_active_subnet(**setting)
        meta["runtime_depth"].append(list(net.runtime_depth))
        net.zero_grad()
        y = net(mg.T(x))
        loss = F.mse_loss(y, mg.T(x))
        loss.backward()
        out["y_s%d" % si] = mg.A(y)
        out["loss_s%d" % si] = np.array(float(loss.detach()))
        names, isnone, s1, sabs, l2 = mg._grad_summary(net)
        out["g_isnone_s%d" % si] = isnone
        out["g_l2_s%d" % si] = l2
    DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = None
    mg.save("x4_net.npz", **out)
    with open(os.path.join(mg.HERE, "x4_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote x4_meta.json")


if __name__ == "__main__":
    main()
