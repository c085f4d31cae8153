"""OFAMobileNetX4 -- learned down-scaler + SR decoder supernet (what train_ofa_net_sr_simple.py builds).

Drop-in for the operator surface of reference ofa/elastic_nn/networks/ofa_mbx4.py:16-453.  As
committed it is an AUTOENCODER (HR -> 2 x [3x3 conv, BN, PixelUnshuffle] -> 4 elastic MB stages ->
3 convs -> 3-channel LR image -> conv -> 4 elastic MB stages -> 2 convs -> 1-2 x [conv, BN,
PixelShuffle] -> conv -> HR; output size == input size).  All convs are 3x3.

blocks[0..1]   conv+BN+PixelUnshuffle      group 0
blocks[2..17]  encoder DynamicMBConvLayer   groups 1..4
blocks[18..33] decoder DynamicMBConvLayer   groups 5..8
blocks[34..35] conv+BN+PixelShuffle         group 9

COMPAT_REFERENCE_INDEXING (default True) reproduces the reference's stage indexing: the depth list becomes
[pd, d0..d6, pd, d7] (two inserts, :362-366) and every stage loop re-enumerates from 0, so un/shuffle stages
read runtime_depth[0] and both encoder stage j and decoder stage j read runtime_depth[j] (:185-254).
False gives the intended mapping (pd for both un/shuffle stages, d[0:4] encoder, d[4:8] decoder).
"""
import random

from ...imagenet_codebase.networks.mobilenet_x4 import MobileNetX4
from ...imagenet_codebase.networks.proxyless_nets import MobileInvertedResidualBlock
from ...layers import ConvLayer, IdentityLayer
from ...utils import int2list
from ..modules.dynamic_layers import DynamicMBConvLayer
from .ofa_mbs4 import OFAMobileNetS4


class OFAMobileNetX4(MobileNetX4):

    COMPAT_REFERENCE_INDEXING = True

    def __init__(self, bn_param=(0.1, 1e-5), dropout_rate=0.1, base_stage_width=None, width_mult_list=1.0,
                 ks_list=3, expand_ratio_list=6, depth_list=4, pixelshuffle_depth_list=2):
        self.width_mult_list = sorted(int2list(width_mult_list, 1))
        self.ks_list = sorted(int2list(ks_list, 1))
        self.expand_ratio_list = sorted(int2list(expand_ratio_list, 1))
        self.depth_list = sorted(int2list(depth_list, 1))
        self.pixelshuffle_depth_list = sorted(int2list(pixelshuffle_depth_list, 1))
        self.base_stage_width = base_stage_width
        max_d, max_pd = max(self.depth_list), max(self.pixelshuffle_depth_list)
        C = 64

        def conv(cin, cout, act=None):
            return ConvLayer(cin, cout, kernel_size=3, stride=1, act_func=act, use_bn=True)

        def mb_stage(blocks, groups):
            groups.append([len(blocks) + i for i in range(max_d)])
            for _ in range(max_d):
                mb = DynamicMBConvLayer(in_channel_list=[C], out_channel_list=[C], kernel_size_list=ks_list,
                                        expand_ratio_list=expand_ratio_list, stride=1, act_func="relu6", use_se=False)
                blocks.append(MobileInvertedResidualBlock(mb, IdentityLayer([C], [C])))

        # encoder: 3 -> 16 ch, unshuffle (x4 channels, /2 resolution), twice
        blocks = [conv(3, 16, "pixelunshuffle"), conv(64, 16, "pixelunshuffle")]
        groups = [[0, 1]]
        for _ in range(4):
            mb_stage(blocks, groups)
        enc_final = [conv(C, C), conv(C, C), conv(C, 3)]
        dec_first = conv(3, C)
        for _ in range(4):
            mb_stage(blocks, groups)
        dec_final = [conv(C, C), conv(C, C)]
        groups.append([len(blocks) + i for i in range(max_pd)])
        for _ in range(max_pd):
            blocks.append(conv(C, 4 * C, "pixelshuffle"))
        head = conv(C, 3)

        self.block_group_info = groups
        self.runtime_depth = [len(g) for g in groups]
        super().__init__(blocks, enc_final, dec_first, dec_final, head, self.runtime_depth)
        self.set_bn_param(momentum=bn_param[0], eps=bn_param[1])

    @staticmethod
    def name():
        return "OFAMobileNetX4"

    def _depth_of(self, group):
        """runtime depth applied to block group `group` (0 unshuffle, 1-4 encoder, 5-8 decoder, 9 shuffle)."""
        if not self.COMPAT_REFERENCE_INDEXING:
            return self.runtime_depth[group]
        if group in (0, 9):
            return self.runtime_depth[0]
        return self.runtime_depth[(group - 1) % 4]

    def _run_groups(self, x, first, last):
        for gidx in range(first, last):
            for idx in self.block_group_info[gidx][:self._depth_of(gidx)]:
                x = self.blocks[idx](x)
        return x

    def forward(self, x):
        x = self._run_groups(x, 0, 1)
        skip = x
        x = self._run_groups(x, 1, 5)
        for i, c in enumerate(self.enc_final_conv_blocks):
            x = c(x)
            if i == 0:
                x = x + skip
        x = self.dec_first_conv_block(x)
        skip = x
        x = self._run_groups(x, 5, 9)
        for i, c in enumerate(self.dec_final_conv_blocks):
            x = c(x)
            if i == 0:
                x = x + skip
        x = self._run_groups(x, 9, 10)
        return self.dec_final_output_conv_block(x)

    def active_block_sequence(self):
        seq = []
        for gidx in range(10):
            kind = "unshuffle" if gidx == 0 else ("shuffle" if gidx == 9 else "mb")
            for idx in self.block_group_info[gidx][:self._depth_of(gidx)]:
                blk = self.blocks[idx]
                seq.append((kind, blk.mobile_inverted_conv if kind == "mb" else blk))
            if gidx == 4:
                seq += [("conv", c) for c in self.enc_final_conv_blocks] + [("conv", self.dec_first_conv_block)]
            if gidx == 8:
                seq += [("conv", c) for c in self.dec_final_conv_blocks]
        seq.append(("head", self.dec_final_output_conv_block))
        return seq

    @property
    def module_str(self):
        lines = []
        for stage, group in enumerate(self.block_group_info):
            lines += [self.blocks[idx].module_str for idx in group[:self.runtime_depth[stage]]]
        lines += [b.module_str for b in self.enc_final_conv_blocks]
        lines.append(self.dec_first_conv_block.module_str)
        lines += [b.module_str for b in self.dec_final_conv_blocks]
        lines.append(self.dec_final_output_conv_block.module_str)
        return "\n".join(lines) + "\n"

    load_weights_from_net = OFAMobileNetS4.load_weights_from_net

    # ------------------------------------------------------------- active sub-network control
    def _mb_blocks(self):
        return self.blocks[2:-len(self.block_group_info[-1])]

    def set_active_subnet(self, wid=None, ks=None, e=None, d=None, pixel_d=None):
        mbs = self._mb_blocks()
        ks = int2list(ks, len(mbs))
        expand = int2list(e, len(mbs))
        depth = int2list(d, len(self.block_group_info) - 2)
        pixel_depth = int2list(pixel_d, 2)
        if self.COMPAT_REFERENCE_INDEXING:
            depth.insert(0, pixel_depth[0])
            depth.insert(-1, pixel_depth[0])    # lands before the last entry (:365), mutating the caller's list
        else:
            depth = [pixel_depth[0]] + list(depth) + [pixel_depth[0]]
        for block, k, ratio in zip(mbs, ks, expand):
            if k is not None:
                block.mobile_inverted_conv.active_kernel_size = k
            if ratio is not None:
                block.mobile_inverted_conv.active_expand_ratio = ratio
        for i, dd in enumerate(depth):
            if dd is not None:
                self.runtime_depth[i] = min(len(self.block_group_info[i]), dd)

    _CONSTRAINT_SLOTS = OFAMobileNetS4._CONSTRAINT_SLOTS
    set_constraint = OFAMobileNetS4.set_constraint
    clear_constraint = OFAMobileNetS4.clear_constraint
    _candidates = OFAMobileNetS4._candidates

    def sample_active_subnet(self):
        """random.choice call order of the reference (:399-453): 32 kernel sizes, 32 expand ratios,
        8 depths, 1 pixel-shuffle depth."""
        n_mb = len(self._mb_blocks())

        def draw(cands, count):
            if not isinstance(cands[0], list):
                cands = [cands for _ in range(count)]
            return [random.choice(c) for c in cands]

        ks_setting = draw(self._candidates("_ks_include_list", self.ks_list), n_mb)
        expand_setting = draw(self._candidates("_expand_include_list", self.expand_ratio_list), n_mb)
        depth_setting = draw(self._candidates("_depth_include_list", self.depth_list), len(self.block_group_info) - 2)
        pd_setting = draw(self._candidates("_pixelshuffle_depth_include_list", self.pixelshuffle_depth_list), 1)
        self.set_active_subnet(None, ks_setting, expand_setting, depth_setting, pd_setting)
        return {"wid": None, "ks": ks_setting, "e": expand_setting, "d": depth_setting, "pixel_d": pd_setting}

    def re_organize_middle_weights(self, expand_ratio_stage=0):
        for block in self._mb_blocks():
            block.mobile_inverted_conv.re_organize_middle_weights(expand_ratio_stage)
