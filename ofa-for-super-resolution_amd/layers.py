"""Static layers of the SR networks (mirror of the SR-relevant part of the reference's
ofa/layers.py): ConvLayer [conv -> BN -> act, where act may be PixelShuffle / PixelUnshuffle on the
HIP reshuffle kernel], IdentityLayer, ZeroLayer, MBInvertedConvLayer (what
DynamicMBConvLayer.get_active_subnet extracts).  Classification-only layers (pooling, linear,
depth-conv, SE, channel shuffle) are out of scope.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .utils import MyModule, build_activation, get_same_padding


def set_layer_from_config(layer_config):
    """reference ofa/layers.py:11-27"""
    if layer_config is None:
        return None
    table = {cls.__name__: cls for cls in (ConvLayer, IdentityLayer, ZeroLayer, MBInvertedConvLayer)}
    cfg = dict(layer_config)
    return table[cfg.pop("name")].build_from_config(cfg)


class My2DLayer(MyModule):
    """weight / bn / act pipeline assembled in `ops_order` (reference ofa/layers.py:30-117).
    Child names ('conv', 'bn', 'act', 'dropout') are the reference's, so state-dict keys match."""

    def __init__(self, in_channels, out_channels, use_bn=True, act_func="relu", dropout_rate=0,
                 ops_order="weight_bn_act"):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.use_bn = use_bn
        self.act_func = act_func
        self.dropout_rate = dropout_rate
        self.ops_order = ops_order

        order = self.ops_list
        if "bn" not in order or "weight" not in order:
            raise ValueError("Invalid ops_order: %s" % ops_order)
        for op in order:
            if op == "weight":
                if dropout_rate > 0:
                    self.add_module("dropout", nn.Dropout2d(dropout_rate, inplace=True))
                for name, mod in (self.weight_op() or {}).items():
                    self.add_module(name, mod)
            elif op == "bn":
                if use_bn:
                    self.add_module("bn", nn.BatchNorm2d(in_channels if self.bn_before_weight else out_channels))
            elif op == "act":
                act = build_activation(act_func, order[0] != "act")
                if act is not None:
                    self.add_module("act", act)

    @property
    def ops_list(self):
        return self.ops_order.split("_")

    @property
    def bn_before_weight(self):
        order = self.ops_list
        return order.index("bn") < order.index("weight")

    def weight_op(self):
        raise NotImplementedError

    def forward(self, x):
        for module in self._modules.values():
            x = module(x)
        return x

    @property
    def config(self):
        return {
            "in_channels": self.in_channels, "out_channels": self.out_channels, "use_bn": self.use_bn,
            "act_func": self.act_func, "dropout_rate": self.dropout_rate, "ops_order": self.ops_order,
        }


class ConvLayer(My2DLayer):
    """reference ofa/layers.py:120-187.  The dense k x k convolution itself runs on MIOpen through
    nn.Conv2d (it is not one of the hot path's named kernels; DESIGN.md 'next' row f1)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, groups=1, bias=False,
                 has_shuffle=False, use_bn=True, act_func="relu", dropout_rate=0, ops_order="weight_bn_act"):
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = dilation
        self.groups = groups
        self.bias = bias
        self.has_shuffle = has_shuffle
        super().__init__(in_channels, out_channels, use_bn, act_func, dropout_rate, ops_order)

    def weight_op(self):
        pad = get_same_padding(self.kernel_size)
        pad = pad * self.dilation if isinstance(pad, int) else (pad[0] * self.dilation, pad[1] * self.dilation)
        if self.has_shuffle and self.groups > 1:
            raise NotImplementedError("channel-shuffle group convs are classification-only (out of scope)")
        return OrderedDict(conv=nn.Conv2d(self.in_channels, self.out_channels, kernel_size=self.kernel_size,
                                          stride=self.stride, padding=pad, dilation=self.dilation,
                                          groups=self.groups, bias=self.bias))

    def forward(self, x):
        # inference (no grad, eval-mode BN, 16-bit activations): conv + BN (+ ReLU6 | PixelShuffle(2)) as ONE kernel with
        # operands prepared once per set of weights (ops.conv_bn_act_infer)
        if (ops.FUSED_INFER and x.is_cuda and not torch.is_grad_enabled() and self.ops_order == "weight_bn_act"
                and self.dropout_rate == 0 and not (self.use_bn and self.bn.training)):
            act = self._modules.get("act", None)
            if act is None:
                code, rest = ops.ACT_NONE, None
            elif self.act_func == "relu6":
                code, rest = ops.ACT_RELU6, None
            elif self.act_func == "pixelshuffle" and getattr(act, "upscale_factor", None) == 2:
                code, rest = ops.ACT_PIXEL_SHUFFLE2, None
            else:
                code, rest = ops.ACT_NONE, act
            y = ops.conv_bn_act_infer(x, self.conv, self.bn if self.use_bn else None, code)
            if y is not None:
                return y if rest is None else rest(y)
        # conv -> fused BatchNorm(+ReLU6) HIP passes -> remaining activation (PixelShuffle ...)
        if not (ops.FUSED_BN and self.ops_order == "weight_bn_act" and self.use_bn and self.dropout_rate == 0
                and x.is_cuda):
            return super().forward(x)
        if self.bn.training:   # BatchNorm statistics from the conv's epilogue, PixelShuffle(2) as the BN apply's store
            act = self._modules.get("act", None)
            if act is None:
                code, rest = ops.ACT_NONE, None
            elif self.act_func == "relu6":
                code, rest = ops.ACT_RELU6, None
            elif self.act_func == "pixelshuffle" and getattr(act, "upscale_factor", None) == 2:
                code, rest = ops.ACT_PIXEL_SHUFFLE2, None
            else:
                code, rest = ops.ACT_NONE, act
            y = ops.conv_bn_act_train(x, self.conv, self.bn, code)
            if y is not None:
                return y if rest is None else rest(y)
        x = ops.conv2d(x, self.conv)
        if self.act_func == "relu6":
            return ops.bn_act(x, self.bn, ops.ACT_RELU6)
        x = ops.bn_act(x, self.bn, ops.ACT_NONE)
        act = self._modules.get("act", None)
        return x if act is None else act(x)

    @property
    def module_str(self):
        ks = (self.kernel_size, self.kernel_size) if isinstance(self.kernel_size, int) else self.kernel_size
        kind = ("Dilated" if self.dilation > 1 else "") + ("Group" if self.groups > 1 else "") + "Conv"
        return "%dx%d_%s_O%d" % (ks[0], ks[1], kind, self.out_channels)

    @property
    def config(self):
        return {
            "name": ConvLayer.__name__, "kernel_size": self.kernel_size, "stride": self.stride,
            "dilation": self.dilation, "groups": self.groups, "bias": self.bias, "has_shuffle": self.has_shuffle,
            **super().config,
        }

    @staticmethod
    def build_from_config(config):
        return ConvLayer(**config)


class IdentityLayer(My2DLayer):
    """reference ofa/layers.py:310-332"""

    def __init__(self, in_channels, out_channels, use_bn=False, act_func=None, dropout_rate=0,
                 ops_order="weight_bn_act"):
        super().__init__(in_channels, out_channels, use_bn, act_func, dropout_rate, ops_order)

    def weight_op(self):
        return None

    @property
    def module_str(self):
        return "Identity"

    @property
    def config(self):
        return {"name": IdentityLayer.__name__, **super().config}

    @staticmethod
    def build_from_config(config):
        return IdentityLayer(**config)


class ZeroLayer(MyModule):
    """placeholder for a removed branch (reference ofa/layers.py ZeroLayer); never executed."""

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        raise ValueError

    @property
    def module_str(self):
        return "Zero"

    @property
    def config(self):
        return {"name": ZeroLayer.__name__, "stride": self.stride}

    @staticmethod
    def build_from_config(config):
        return ZeroLayer(**config)


class _DepthwiseConv2d(nn.Conv2d):
    """nn.Conv2d(groups=C) whose forward is the HIP depthwise kernel (parameter name/shape unchanged)."""

    def forward(self, x):
        k = self.kernel_size[0]
        if (self.stride != (1, 1) or self.dilation != (1, 1) or self.groups != self.in_channels
                or self.kernel_size[0] != self.kernel_size[1] or self.bias is not None):
            raise NotImplementedError("only stride-1 square depthwise convs are on the SR hot path")
        return ops.dwconv(x, self.weight)


class _PointwiseConv2d(nn.Conv2d):
    """nn.Conv2d 1x1 whose forward is the HIP MFMA pointwise kernel."""

    def forward(self, x):
        return ops.pwconv(x, self.weight, self.out_channels)


class MBInvertedConvLayer(MyModule):
    """Fixed-architecture inverted bottleneck: the extraction target of
    DynamicMBConvLayer.get_active_subnet (reference ofa/layers.py:447-527); runs on the same HIP
    kernels as the elastic layer."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, expand_ratio=6, mid_channels=None,
                 act_func="relu6", use_se=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.expand_ratio = expand_ratio
        self.mid_channels = mid_channels
        self.act_func = act_func
        self.use_se = use_se
        if use_se:
            raise NotImplementedError("squeeze-excite is classification-only (se_stages all False in the SR nets)")

        feature_dim = round(in_channels * expand_ratio) if mid_channels is None else mid_channels
        if expand_ratio == 1:
            self.inverted_bottleneck = None
        else:
            self.inverted_bottleneck = nn.Sequential(OrderedDict([
                ("conv", _PointwiseConv2d(in_channels, feature_dim, 1, 1, 0, bias=False)),
                ("bn", nn.BatchNorm2d(feature_dim)),
                ("act", build_activation(act_func, inplace=True)),
            ]))
        self.depth_conv = nn.Sequential(OrderedDict([
            ("conv", _DepthwiseConv2d(feature_dim, feature_dim, kernel_size, stride, get_same_padding(kernel_size),
                                      groups=feature_dim, bias=False)),
            ("bn", nn.BatchNorm2d(feature_dim)),
            ("act", build_activation(act_func, inplace=True)),
        ]))
        self.point_linear = nn.Sequential(OrderedDict([
            ("conv", _PointwiseConv2d(feature_dim, out_channels, 1, 1, 0, bias=False)),
            ("bn", nn.BatchNorm2d(out_channels)),
        ]))

    def forward(self, x):
        if self.inverted_bottleneck is not None:
            x = self.inverted_bottleneck(x)
        return self.point_linear(self.depth_conv(x))

    @property
    def module_str(self):
        ratio = self.expand_ratio if self.mid_channels is None else self.mid_channels // self.in_channels
        return "%dx%d_MBConv%d_%s_O%d" % (self.kernel_size, self.kernel_size, ratio, self.act_func.upper(),
                                          self.out_channels)

    @property
    def config(self):
        return {
            "name": MBInvertedConvLayer.__name__, "in_channels": self.in_channels,
            "out_channels": self.out_channels, "kernel_size": self.kernel_size, "stride": self.stride,
            "expand_ratio": self.expand_ratio, "mid_channels": self.mid_channels, "act_func": self.act_func,
            "use_se": self.use_se,
        }

    @staticmethod
    def build_from_config(config):
        return MBInvertedConvLayer(**config)
