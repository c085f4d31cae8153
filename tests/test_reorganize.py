"""re_organize_middle_weights (DynamicMBConvLayer, reference dynamic_layers.py:156-199; net-level driver
ofa_mbs4.py:462-464, used by supporting_elastic_expand, progressive_shrinking.py:331-396) against the reference's own
result on det-filled weights (tests/golden/reorganize.npz, make_golden.py gen_reorganize).  Pure index permutations of
fp32 tensors: bit-exact.  CPU-only (no kernel involved)."""
import numpy as np
import torch

from conftest import amd
from detfill import fill_state_dict


def _load(module, prefix):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(shapes, prefix).items()})


def test_layer_reorganize_matches_reference(golden):
    g = golden("reorganize.npz")
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    layer = amd("elastic_nn.modules.dynamic_layers").DynamicMBConvLayer([16], [16], [3, 5, 7], [3, 4, 6], stride=1,
                                                                        act_func="relu6")
    _load(layer, "reorg")
    for stage in (0, 1, 2):
        layer.re_organize_middle_weights(expand_ratio_stage=stage)
        sd = layer.state_dict()
        keys = [k for k in g.files if k.startswith("layer_s%d_" % stage)]
        assert len(keys) == len(sd)
        for k in keys:
            assert np.array_equal(sd[k[len("layer_s%d_" % stage):]].numpy(), g[k]), (stage, k)


def test_net_reorganize_matches_reference(golden):
    g = golden("reorganize.npz")
    dop = amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = amd("elastic_nn.networks").OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6],
                                                    depth_list=[2, 3, 4], pixelshuffle_depth_list=[1, 2])
    _load(net, "s4")
    for stage in (0, 1):
        net.re_organize_middle_weights(expand_ratio_stage=stage)
        sd = net.state_dict()
        keys = [k for k in g.files if k.startswith("net_s%d_" % stage)]
        assert len(keys) > 30
        for k in keys:
            assert np.array_equal(sd[k[len("net_s%d_" % stage):]].numpy(), g[k]), (stage, k)
