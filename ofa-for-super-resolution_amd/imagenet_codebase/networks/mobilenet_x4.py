"""MobileNetX4: container base of the learned-downscale + SR autoencoder (reference
ofa/imagenet_codebase/networks/mobilenet_x4.py:14-27); attribute names fix the state-dict keys."""
import torch.nn as nn

from ...utils import MyNetwork


class MobileNetX4(MyNetwork):

    def __init__(self, blocks, enc_final_conv_blocks, dec_first_conv_block, dec_final_conv_blocks,
                 dec_final_output_conv_block, runtime_depth):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        self.enc_final_conv_blocks = nn.ModuleList(enc_final_conv_blocks)
        self.dec_first_conv_block = dec_first_conv_block
        self.dec_final_conv_blocks = nn.ModuleList(dec_final_conv_blocks)
        self.dec_final_output_conv_block = dec_final_output_conv_block
        self.runtime_depth = runtime_depth

    def forward(self, x):
        return x
