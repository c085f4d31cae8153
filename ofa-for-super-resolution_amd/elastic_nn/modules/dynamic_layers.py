"""DynamicMBConvLayer -- the elastic inverted-bottleneck block of the OFA-SR supernet, on the HIP
operators of dynamic_op.py.

Drop-in for reference ofa/elastic_nn/modules/dynamic_layers.py:14-199: same constructor, same
sub-module / parameter names (=> same state-dict keys), same `active_*` attribute protocol,
`get_active_subnet` and `re_organize_middle_weights`.  DynamicConvLayer / DynamicLinearLayer are
unused by the SR nets and out of scope.
"""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from ... import ops
from ...layers import MBInvertedConvLayer
from ...utils import MyModule, build_activation, get_net_device, int2list, make_divisible
from ..utils import adjust_bn_according_to_idx, copy_bn
from .dynamic_op import DynamicBatchNorm2d, DynamicPointConv2d, DynamicSeparableConv2d


def _conv_bn_act(conv, bn, act_func):
    mods = [("conv", conv), ("bn", bn)]
    act = build_activation(act_func, inplace=True)
    if act is not None:
        mods.append(("act", act))
    return nn.Sequential(OrderedDict(mods))


class DynamicMBConvLayer(MyModule):
    accepts_residual = True
    """expand 1x1 (C_in -> mid) -> BN -> act -> depthwise kxk -> BN -> act -> project 1x1 (mid -> C_out) -> BN
    with mid = make_divisible(round(C_in * active_expand_ratio), 8) and k = active_kernel_size."""

    def __init__(self, in_channel_list, out_channel_list, kernel_size_list=3, expand_ratio_list=6, stride=1,
                 act_func="relu6", use_se=False):
        super().__init__()
        self.in_channel_list = in_channel_list
        self.out_channel_list = out_channel_list
        self.kernel_size_list = int2list(kernel_size_list, 1)
        self.expand_ratio_list = int2list(expand_ratio_list, 1)
        self.stride = stride
        self.act_func = act_func
        self.use_se = use_se
        if use_se:
            raise NotImplementedError("squeeze-excite is classification-only (se_stages all False in the SR nets)")

        c_in, c_out = max(self.in_channel_list), max(self.out_channel_list)
        mid_max = round(c_in * max(self.expand_ratio_list))
        if max(self.expand_ratio_list) == 1:
            self.inverted_bottleneck = None
        else:
            self.inverted_bottleneck = _conv_bn_act(DynamicPointConv2d(c_in, mid_max), DynamicBatchNorm2d(mid_max),
                                                    act_func)
        self.depth_conv = _conv_bn_act(DynamicSeparableConv2d(mid_max, self.kernel_size_list, stride),
                                       DynamicBatchNorm2d(mid_max), act_func)
        self.point_linear = _conv_bn_act(DynamicPointConv2d(mid_max, c_out), DynamicBatchNorm2d(c_out), None)

        self.active_kernel_size = max(self.kernel_size_list)
        self.active_expand_ratio = max(self.expand_ratio_list)
        self.active_out_channel = c_out

    def active_middle_channel(self, in_channel):
        return make_divisible(round(in_channel * self.active_expand_ratio), 8)

    def forward(self, x, residual=None):
        """`residual` (optional) is added to the block output inside the last BN pass -- what
        MobileInvertedResidualBlock does with its identity shortcut (reference proxyless_nets.py:50)."""
        if self.inverted_bottleneck is not None:
            self.inverted_bottleneck.conv.active_out_channel = self.active_middle_channel(x.size(1))
        self.depth_conv.conv.active_kernel_size = self.active_kernel_size
        self.point_linear.conv.active_out_channel = self.active_out_channel
        fused = ops.FUSED_BN and self.act_func == "relu6" and not DynamicBatchNorm2d.SET_RUNNING_STATISTICS
        if self.composite_eligible(x):
            return self._forward_composite(x, residual)
        if not fused:
            if self.inverted_bottleneck is not None:
                x = self.inverted_bottleneck(x)
            x = self.point_linear(self.depth_conv(x))
            return x if residual is None else x + residual
        # conv -> [BN + ReLU6] fused passes; the last BN also folds in the shortcut add
        if self.inverted_bottleneck is not None:
            x = ops.bn_act(self.inverted_bottleneck.conv(x), self.inverted_bottleneck.bn.bn, ops.ACT_RELU6)
        x = ops.bn_act(self.depth_conv.conv(x), self.depth_conv.bn.bn, ops.ACT_RELU6)
        return ops.bn_act(self.point_linear.conv(x), self.point_linear.bn.bn, ops.ACT_NONE, residual)

    def composite_eligible(self, x):
        """the composite HIP call (one per block, ops.FusedMBConvFn, or one per stack, ops.FusedMBStackFn) serves this block"""
        fused = ops.FUSED_BN and self.act_func == "relu6" and not DynamicBatchNorm2d.SET_RUNNING_STATISTICS
        return (fused and ops.FUSED_BLOCK and self.inverted_bottleneck is not None and x.is_cuda and self.stride == 1
                and all(m.bn.momentum is not None for m in (self.inverted_bottleneck.bn, self.depth_conv.bn,
                                                            self.point_linear.bn)))

    def composite_args(self, in_channels, add_x):
        """(cfg, params) of the composite call for the ACTIVE sub-network: cfg as ops.FusedMBConvFn / FusedMBStackFn read
        it, params = (w1, g1, b1, wdw, g2, b2, w2, g3, b3, *transform matrices walked)"""
        dw = self.depth_conv.conv
        K = self.active_kernel_size
        chain = dw._chain(K)
        transform = dw.KERNEL_TRANSFORM_MODE is not None and K < max(dw.kernel_size_list)
        mats = [getattr(dw, "%dto%d_matrix" % (a, b)) for a, b in zip(chain[:-1], chain[1:])] if transform else []
        bn1, bn2, bn3 = self.inverted_bottleneck.bn.bn, self.depth_conv.bn.bn, self.point_linear.bn.bn
        cfg = {"mid": self.active_middle_channel(in_channels), "out": self.active_out_channel, "K": K, "chain": chain,
               "residual": add_x, "bns": (bn1, bn2, bn3), "owner": self, "nparams": 9 + len(mats)}
        params = (self.inverted_bottleneck.conv.conv.weight, bn1.weight, bn1.bias, dw.conv.weight, bn2.weight, bn2.bias,
                  self.point_linear.conv.conv.weight, bn3.weight, bn3.bias) + tuple(mats)
        return cfg, params

    def _forward_composite(self, x, residual):
        """the whole block (+ shortcut when `residual is x`) as one composite HIP call (ops.FusedMBConvFn)."""
        add_x = residual is not None and residual is x
        cfg, params = self.composite_args(x.size(1), add_x)
        bn1, bn2, bn3 = cfg["bns"]
        if ops.FUSED_INFER and not torch.is_grad_enabled() and not (bn1.training or bn2.training or bn3.training) \
                and (add_x or residual is None):
            # inference with eval-mode BN: the block is one kernel (BN folded, the mid tensor never reaches HBM)
            y = ops.mbconv_infer(x, cfg, *params)
            if y is not None:
                return y
        y = ops.FusedMBConvFn.apply(x, cfg, *params)
        if residual is not None and not add_x:
            y = y + residual
        return y

    @property
    def module_str(self):
        return "(O%d, E%.1f, K%d)" % (self.active_out_channel, self.active_expand_ratio, self.active_kernel_size)

    @property
    def config(self):
        return {
            "name": DynamicMBConvLayer.__name__, "in_channel_list": self.in_channel_list,
            "out_channel_list": self.out_channel_list, "kernel_size_list": self.kernel_size_list,
            "expand_ratio_list": self.expand_ratio_list, "stride": self.stride, "act_func": self.act_func,
            "use_se": self.use_se,
        }

    @staticmethod
    def build_from_config(config):
        return DynamicMBConvLayer(**config)

    # ------------------------------------------------------------------ sub-network extraction
    def get_active_subnet(self, in_channel, preserve_weight=True):
        """a static MBInvertedConvLayer holding the active slice of every weight (reference :112-154)."""
        mid = self.active_middle_channel(in_channel)
        sub = MBInvertedConvLayer(in_channel, self.active_out_channel, self.active_kernel_size, self.stride,
                                  self.active_expand_ratio, act_func=self.act_func, mid_channels=mid,
                                  use_se=self.use_se).to(get_net_device(self))
        if not preserve_weight:
            return sub
        with torch.no_grad():
            if sub.inverted_bottleneck is not None:
                sub.inverted_bottleneck.conv.weight.copy_(
                    self.inverted_bottleneck.conv.conv.weight[:mid, :in_channel])
                copy_bn(sub.inverted_bottleneck.bn, self.inverted_bottleneck.bn.bn)
            sub.depth_conv.conv.weight.copy_(self.depth_conv.conv.get_active_filter(mid, self.active_kernel_size))
            copy_bn(sub.depth_conv.bn, self.depth_conv.bn.bn)
            sub.point_linear.conv.weight.copy_(self.point_linear.conv.conv.weight[:self.active_out_channel, :mid])
            copy_bn(sub.point_linear.bn, self.point_linear.bn.bn)
        return sub

    def re_organize_middle_weights(self, expand_ratio_stage=0):
        """sort the middle channels by the L1 importance of the project weights so that narrower
        expand ratios keep the most important channels (reference :156-199)."""
        ops.clear_infer_cache()   # parameters are replaced through .data below
        w_proj = self.point_linear.conv.conv.weight.data
        importance = torch.sum(torch.abs(w_proj), dim=(0, 2, 3))
        if expand_ratio_stage > 0:
            widths = sorted(copy.deepcopy(self.expand_ratio_list), reverse=True)
            keep = round(max(self.in_channel_list) * widths[expand_ratio_stage])
            # channels beyond the already-shrunk width keep their order, below everything else
            importance[keep:] = torch.arange(0, keep - importance.size(0), -1, device=importance.device,
                                             dtype=importance.dtype)
        _, order = torch.sort(importance, dim=0, descending=True)
        self.point_linear.conv.conv.weight.data = torch.index_select(w_proj, 1, order)
        adjust_bn_according_to_idx(self.depth_conv.bn.bn, order)
        self.depth_conv.conv.conv.weight.data = torch.index_select(self.depth_conv.conv.conv.weight.data, 0, order)
        if self.inverted_bottleneck is None:
            return order
        adjust_bn_according_to_idx(self.inverted_bottleneck.bn.bn, order)
        self.inverted_bottleneck.conv.conv.weight.data = torch.index_select(
            self.inverted_bottleneck.conv.conv.weight.data, 0, order)
        return None
