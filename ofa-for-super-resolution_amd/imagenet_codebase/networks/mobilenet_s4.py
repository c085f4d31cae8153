"""MobileNetS4: container base of the SR-only network (reference
ofa/imagenet_codebase/networks/mobilenet_s4.py:15-31).  Attribute names are the reference's, so
state-dict keys match (`blocks.N...`, `dec_first_conv_block...`, `dec_final_conv_blocks.N...`,
`dec_final_output_conv_block...`)."""
import torch.nn as nn

from ...layers import IdentityLayer, MBInvertedConvLayer
from ...utils import MyNetwork
from .proxyless_nets import MobileInvertedResidualBlock


class MobileNetS4(MyNetwork):

    def __init__(self, blocks, dec_first_conv_block, dec_final_conv_blocks, dec_final_output_conv_block,
                 runtime_depth):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        self.dec_first_conv_block = dec_first_conv_block
        self.dec_final_conv_blocks = nn.ModuleList(dec_final_conv_blocks)
        self.dec_final_output_conv_block = dec_final_output_conv_block
        self.runtime_depth = runtime_depth

    def forward(self, x):
        return x

    def zero_last_gamma(self):
        from ... import ops
        ops.clear_infer_cache()
        for m in self.modules():
            if isinstance(m, MobileInvertedResidualBlock) and isinstance(m.mobile_inverted_conv, MBInvertedConvLayer) \
                    and isinstance(m.shortcut, IdentityLayer):
                m.mobile_inverted_conv.point_linear.bn.weight.data.zero_()
