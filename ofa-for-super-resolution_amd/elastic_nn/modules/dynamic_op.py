"""Elastic operators of the OFA-SR supernet on MI355X HIP kernels.

Drop-in for the reference's ofa/elastic_nn/modules/dynamic_op.py (same class names, constructor
signatures, parameter names -> same state-dict keys, same attribute-mutation API), with every
F.conv2d / F.linear call site replaced by a hand-written gfx950 kernel behind the C ABI
(include/ofasr.h):

  DynamicSeparableConv2d  reference :14-84   -> ofasr_ktransform_* + ofasr_dwconv_*
  DynamicPointConv2d      reference :87-112  -> ofasr_pwconv_*   (weight slice read in place)
  DynamicBatchNorm2d      reference :139-172 -> sliced BatchNorm (ATen on the GPU; BN is not one of
                                                the named kernels -- DESIGN.md)
DynamicLinear / DynamicSE are classification-only (SR nets have no classifier, se_stages all
False: reference ofa_mbs4.py:50) and are out of scope.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from ... import ops
from ...utils import get_same_padding, sub_filter_start_end


class DynamicSeparableConv2d(nn.Module):
    """Depthwise conv with elastic kernel size.

    `conv.weight` is the max-size [C_max, 1, k_max, k_max] filter; smaller kernels are its centre
    crop, optionally passed through the learned '%dto%d_matrix' chain.  The transform parameters are
    registered only if KERNEL_TRANSFORM_MODE is set AT CONSTRUCTION TIME (reference :32-42)."""

    KERNEL_TRANSFORM_MODE = None  # None or 1

    def __init__(self, max_in_channels, kernel_size_list, stride=1, dilation=1):
        super().__init__()
        self.max_in_channels = max_in_channels
        self.kernel_size_list = kernel_size_list
        self.stride = stride
        self.dilation = dilation

        self.conv = nn.Conv2d(max_in_channels, max_in_channels, max(kernel_size_list), stride,
                              groups=max_in_channels, bias=False)
        self._ks_set = sorted(set(kernel_size_list))
        if self.KERNEL_TRANSFORM_MODE is not None:
            for small, large in zip(self._ks_set[:-1], self._ks_set[1:]):
                self.register_parameter("%dto%d_matrix" % (large, small), Parameter(torch.eye(small ** 2)))
        self.active_kernel_size = max(kernel_size_list)

    # -- the chain of kernel sizes walked from the max kernel down to `kernel_size` (reference :54-69)
    def _chain(self, kernel_size):
        chain = [k for k in reversed(self._ks_set) if k >= kernel_size]
        if not chain or chain[-1] != kernel_size:
            # a size outside kernel_size_list: the reference crops straight from the max kernel
            chain = [self._ks_set[-1], kernel_size] if kernel_size != self._ks_set[-1] else [kernel_size]
        return tuple(chain)

    def get_active_filter(self, in_channel, kernel_size):
        """[in_channel, 1, k, k] fp32 filter of the active sub-kernel -- one HIP launch."""
        transform = self.KERNEL_TRANSFORM_MODE is not None and kernel_size < max(self.kernel_size_list)
        chain = self._chain(kernel_size)
        mats = []
        if transform:
            mats = [getattr(self, "%dto%d_matrix" % (a, b)) for a, b in zip(chain[:-1], chain[1:])]
        return ops.KTransformFn.apply(self.conv.weight, in_channel, chain, transform, *mats)

    def forward(self, x, kernel_size=None):
        if kernel_size is None:
            kernel_size = self.active_kernel_size
        if self.stride != 1 or self.dilation != 1:
            raise NotImplementedError("the SR supernets run every depthwise conv at stride 1, dilation 1 "
                                      "(reference ofa_mbs4.py:48); other strides are out of the hot path")
        get_same_padding(kernel_size)  # asserts an odd kernel, like the reference
        filters = self.get_active_filter(x.size(1), kernel_size)
        return ops.dwconv(x, filters)


class DynamicPointConv2d(nn.Module):
    """1x1 conv on the [:out, :in] slice of a max-size weight, read in place (reference :87-112)."""

    def __init__(self, max_in_channels, max_out_channels, kernel_size=1, stride=1, dilation=1):
        super().__init__()
        self.max_in_channels = max_in_channels
        self.max_out_channels = max_out_channels
        self.kernel_size = kernel_size
        self.stride = stride
        self.dilation = dilation
        self.conv = nn.Conv2d(max_in_channels, max_out_channels, kernel_size, stride=stride, bias=False)
        self.active_out_channel = max_out_channels

    def forward(self, x, out_channel=None):
        if out_channel is None:
            out_channel = self.active_out_channel
        if self.kernel_size != 1 or self.stride != 1:
            raise NotImplementedError("DynamicPointConv2d is used with kernel_size=1, stride=1 in the SR path")
        return ops.pwconv(x, self.conv.weight, out_channel)


class DynamicBatchNorm2d(nn.Module):
    """BatchNorm2d over the first x.size(1) channels of max-size parameters / buffers
    (reference :139-172, including the manual num_batches_tracked bump of the sliced path)."""

    SET_RUNNING_STATISTICS = False

    def __init__(self, max_feature_dim):
        super().__init__()
        self.max_feature_dim = max_feature_dim
        self.bn = nn.BatchNorm2d(max_feature_dim)

    @staticmethod
    def bn_forward(x, bn, feature_dim):
        if bn.num_features == feature_dim or DynamicBatchNorm2d.SET_RUNNING_STATISTICS:
            return bn(x)
        factor = 0.0
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
            factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
        return F.batch_norm(
            x, bn.running_mean[:feature_dim], bn.running_var[:feature_dim], bn.weight[:feature_dim],
            bn.bias[:feature_dim], bn.training or not bn.track_running_stats, factor, bn.eps)

    def forward(self, x):
        if ops.FUSED_BN and not DynamicBatchNorm2d.SET_RUNNING_STATISTICS:
            return ops.bn_act(x, self.bn, ops.ACT_NONE)      # HIP statistics + apply passes
        return self.bn_forward(x, self.bn, x.size(1))
