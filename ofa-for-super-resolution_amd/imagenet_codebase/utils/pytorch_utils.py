"""Parameter / FLOPs accounting used for net_info.txt (reference
ofa/imagenet_codebase/utils/pytorch_utils.py:72-124,190-228).  Closed-form MAC count of the conv
layers of the ACTIVE sub-network, evaluated from shapes only (no forward pass, no hooks)."""
import torch.nn as nn


def count_parameters(net):
    return sum(p.numel() for p in net.parameters() if p.requires_grad)


def count_net_flops(net, data_shape=(1, 3, 64, 64)):
    """MACs of every convolution on the active path of an OFAMobileNetS4-style net for one input of
    `data_shape` (reference counts MACs, convs only).  Returns the count as a float."""
    from ...elastic_nn.modules.dynamic_layers import DynamicMBConvLayer
    from ...layers import ConvLayer

    _, _, h, w = data_shape
    total = 0.0

    def conv_macs(layer, hh, ww):
        c = layer.conv
        return (c.in_channels // c.groups) * c.out_channels * c.kernel_size[0] * c.kernel_size[1] * hh * ww

    blocks = getattr(net, "active_block_sequence", None)
    if blocks is None:
        raise NotImplementedError("count_net_flops needs a net exposing active_block_sequence()")
    for kind, mod in blocks():
        if isinstance(mod, ConvLayer):
            total += conv_macs(mod, h, w)
            if mod.act_func is not None and "pixelshuffle" in mod.act_func and "unshuffle" not in mod.act_func:
                h, w = h * 2, w * 2
            elif mod.act_func is not None and "pixelunshuffle" in mod.act_func:
                h, w = h // 2, w // 2
        elif isinstance(mod, DynamicMBConvLayer):
            cin = max(mod.in_channel_list)
            mid = mod.active_middle_channel(cin)
            k = mod.active_kernel_size
            total += (cin * mid + mid * k * k + mid * mod.active_out_channel) * h * w
    return total


def get_net_info(net, input_shape=(3, 64, 64), measure_latency=None, print_info=True, args=None):
    """reference :190-228 (latency measurement call sites are commented out there)."""
    info = {"params": count_parameters(net)}
    try:
        info["flops"] = count_net_flops(net, (1,) + tuple(input_shape))
    except NotImplementedError:
        info["flops"] = None
    if print_info:
        print("Total training params: %.2fM" % (info["params"] / 1e6))
        if info["flops"] is not None:
            print("Total MACs: %.1fM" % (info["flops"] / 1e6))
    return info
