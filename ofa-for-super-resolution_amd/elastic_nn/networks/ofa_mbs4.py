"""OFAMobileNetS4 -- the SR-only once-for-all supernet (2x / 4x) on the MI355X hot-path kernels.

Drop-in for reference ofa/elastic_nn/networks/ofa_mbs4.py:16-370:

    first 5x5 conv (3 -> 64) .............................. dec_first_conv_block
    4 elastic stages x up to 4 DynamicMBConvLayer(64 -> 64) blocks[0..15]   (residual)
    2 static 5x5 convs (long skip added after the first) .. dec_final_conv_blocks
    1-2 x [5x5 conv 64 -> 256, BN, PixelShuffle(2)] ....... blocks[16..17]
    final 5x5 conv (64 -> 3) + BN ......................... dec_final_output_conv_block

Elastic dimensions: kernel size {3,5,7}, expand ratio {3,4,6}, depth per stage {2,3,4} and the
number of PixelShuffle stages {1,2}.

Stage-indexing compatibility.  As committed, the reference
  * inserts `pixel_d` BEFORE the last entry of the depth list (so it lands on the 4th MB stage and
    the caller's list is mutated), and
  * gates the shuffle stage with runtime_depth[0] (it enumerates a slice).
`COMPAT_REFERENCE_INDEXING = True` (default) reproduces both exactly -- required for parity with
reference checkpoints / logs; set it to False for the evidently intended behaviour (pixel_d
drives the shuffle stage, d the four MB stages).
"""
import random

import torch

from ... import ops
from ...imagenet_codebase.networks.mobilenet_s4 import MobileNetS4
from ...imagenet_codebase.networks.proxyless_nets import MobileInvertedResidualBlock
from ...layers import ConvLayer, IdentityLayer
from ...utils import int2list, make_divisible
from ..modules.dynamic_layers import DynamicMBConvLayer

_N_MB_STAGES = 4


class OFAMobileNetS4(MobileNetS4):

    COMPAT_REFERENCE_INDEXING = True

    def __init__(self, bn_param=(0.1, 1e-5), dropout_rate=0.1, base_stage_width=None, width_mult_list=1.0,
                 ks_list=7, expand_ratio_list=6, depth_list=4, pixelshuffle_depth_list=2):
        self.width_mult_list = sorted(int2list(width_mult_list, 1))
        self.ks_list = sorted(int2list(ks_list, 1))
        self.expand_ratio_list = sorted(int2list(expand_ratio_list, 1))
        self.depth_list = sorted(int2list(depth_list, 1))
        self.pixelshuffle_depth_list = sorted(int2list(pixelshuffle_depth_list, 1))
        self.base_stage_width = base_stage_width

        # fixed widths (reference :36): stem 64 | 4 MB stages 64 | 2 res convs 64 | shuffle 256 | RGB 3
        widths = [[make_divisible(w * m, 1) for m in self.width_mult_list]
                  for w in (64, 64, 64, 64, 64, 64, 64, 256, 3)]
        max_d, max_pd = max(self.depth_list), max(self.pixelshuffle_depth_list)

        stem = ConvLayer(3, max(widths[0]), kernel_size=5, stride=1, act_func=None, use_bn=True)

        blocks, groups = [], []
        feat = widths[0]
        for stage in range(_N_MB_STAGES):
            out = widths[1 + stage]
            groups.append([len(blocks) + i for i in range(max_d)])
            for _ in range(max_d):
                mb = DynamicMBConvLayer(in_channel_list=feat, out_channel_list=out, kernel_size_list=ks_list,
                                        expand_ratio_list=expand_ratio_list, stride=1, act_func="relu6",
                                        use_se=False)
                blocks.append(MobileInvertedResidualBlock(mb, IdentityLayer(feat, feat)))
                feat = out

        res_convs = []
        for out in widths[5:7]:
            res_convs.append(ConvLayer(max(feat), max(out), kernel_size=5, stride=1, act_func=None, use_bn=True))
            feat = out

        groups.append([len(blocks) + i for i in range(max_pd)])
        for _ in range(max_pd):
            # 64 -> 256 channels, PixelShuffle(2) folds them back to 64 at twice the resolution
            blocks.append(ConvLayer(max(feat), max(widths[7]), kernel_size=5, stride=1, act_func="pixelshuffle",
                                    use_bn=True))

        head = ConvLayer(max(feat), max(widths[8]), kernel_size=5, stride=1, act_func=None, use_bn=True)

        self.block_group_info = groups
        self.runtime_depth = [len(g) for g in groups]
        super().__init__(blocks, stem, res_convs, head, self.runtime_depth)
        self.set_bn_param(momentum=bn_param[0], eps=bn_param[1])

    @staticmethod
    def name():
        return "OFAMobileNetS4"

    # ----------------------------------------------------------------------------- forward
    def _shuffle_depth(self):
        return self.runtime_depth[0] if self.COMPAT_REFERENCE_INDEXING else self.runtime_depth[_N_MB_STAGES]

    def active_upscale(self):
        """2 ** (number of conv+PixelShuffle blocks on the active path): the LR input a training / validation step
        must feed.  Under COMPAT_REFERENCE_INDEXING this is NOT 2 ** pixel_d (quirk Q1: the shuffle stage reads
        runtime_depth[0]), so callers ask the net instead of trusting the sampled pixel_d."""
        return 2 ** len(self.block_group_info[_N_MB_STAGES][:self._shuffle_depth()])

    def active_block_sequence(self):
        """(kind, module) for every module on the active path, in execution order."""
        seq = [("stem", self.dec_first_conv_block)]
        for stage in range(_N_MB_STAGES):
            for idx in self.block_group_info[stage][:self.runtime_depth[stage]]:
                seq.append(("mb", self.blocks[idx].mobile_inverted_conv))
        seq += [("res", m) for m in self.dec_final_conv_blocks]
        for idx in self.block_group_info[_N_MB_STAGES][:self._shuffle_depth()]:
            seq.append(("shuffle", self.blocks[idx]))
        seq.append(("head", self.dec_final_output_conv_block))
        return seq

    def _mb_stack(self, x):
        """the active MB blocks, stage by stage (reference ofa_mbs4.py:147-151).  On the GPU with the composite path on
        and gradients (or train-mode BN) in play they run as ONE autograd node / foreign call (ops.FusedMBStackFn); the
        one-kernel inference blocks and every other configuration go block by block."""
        active = [self.blocks[idx] for stage in range(_N_MB_STAGES)
                  for idx in self.block_group_info[stage][:self.runtime_depth[stage]]]
        infer = ops.FUSED_INFER and not torch.is_grad_enabled() and not self.training
        if (ops.FUSED_STACK and active and x.is_cuda and not infer and all(b.stackable(x) for b in active)):
            if torch.is_autocast_enabled() and x.dtype == torch.float32:
                x = x.to(torch.get_autocast_dtype("cuda"))
            cfgs, params, ch = [], [], x.size(1)
            for b in active:
                mb = b.mobile_inverted_conv
                # what DynamicMBConvLayer.forward sets on its children before it runs them
                mb.inverted_bottleneck.conv.active_out_channel = mb.active_middle_channel(ch)
                mb.depth_conv.conv.active_kernel_size = mb.active_kernel_size
                mb.point_linear.conv.active_out_channel = mb.active_out_channel
                cfg, ps = mb.composite_args(ch, True)
                cfgs.append(cfg)
                params.extend(ps)
                ch = mb.active_out_channel
            return ops.mbstack(x, cfgs, params)
        for b in active:
            x = b(x)
        return x

    def forward(self, x):
        with ops.batched_counters():
            return self._forward(x)

    def _forward(self, x):
        x = self.dec_first_conv_block(x)
        skip = x
        x = self._mb_stack(x)
        # everything below is the decoder tail: in backward its gradients are complete when the pass comes back here
        x = ops.grad_milestone(x, "decoder_tail")
        for i, conv in enumerate(self.dec_final_conv_blocks):
            x = conv(x)
            if i == 0:
                x = x + skip
        for idx in self.block_group_info[_N_MB_STAGES][:self._shuffle_depth()]:
            x = self.blocks[idx](x)
        return self.dec_final_output_conv_block(x)

    def early_gradient_parameters(self):
        """the parameters behind ops.grad_milestone(..., "decoder_tail"): residual convs, conv + PixelShuffle blocks, RGB
        head -- their gradients are final first in a backward pass (distributed.FlatGradReducer(early_params=...))"""
        mods = list(self.dec_final_conv_blocks) + [self.blocks[i] for i in self.block_group_info[_N_MB_STAGES]] + \
            [self.dec_final_output_conv_block]
        return [p for m in mods for p in m.parameters()]

    @property
    def module_str(self):
        lines = []
        for stage, group in enumerate(self.block_group_info):
            depth = self.runtime_depth[stage]
            lines += [self.blocks[idx].module_str for idx in group[:depth]]
        lines.append(self.dec_first_conv_block.module_str)
        lines += [b.module_str for b in self.dec_final_conv_blocks]
        lines.append(self.dec_final_output_conv_block.module_str)
        return "\n".join(lines) + "\n"

    @property
    def config(self):
        return {
            "name": OFAMobileNetS4.__name__,
            "bn": self.get_bn_param(),
            "blocks": [b.config for b in self.blocks],
            "dec_first_conv_block": self.dec_first_conv_block.config,
            "dec_final_conv_blocks": [b.config for b in self.dec_final_conv_blocks],
            "dec_final_output_conv_block": self.dec_final_output_conv_block.config,
        }

    @staticmethod
    def build_from_config(config):
        raise ValueError("do not support this function")

    # ------------------------------------------------------------------ checkpoint interchange
    def load_weights_from_net(self, src_model_dict):
        """load a state dict saved from a static net, a DataParallel-wrapped net or another
        supernet: strips 'module.' and maps static <-> dynamic key spellings
        ('.conv.weight' <-> '.conv.conv.weight', '.bn.' <-> '.bn.bn.') -- reference :221-259."""
        own = self.state_dict()
        for raw_key, value in src_model_dict.items():
            key = raw_key.replace("module.", "")
            if key in own:
                new_key = key
            elif ".bn.bn." in key:
                new_key = key.replace(".bn.bn.", ".bn.")
            elif ".conv.conv.weight" in key:
                new_key = key.replace(".conv.conv.weight", ".conv.weight")
            elif "bn." in key:
                new_key = key.replace("bn.", "bn.bn.")
            elif "conv.weight" in key:
                new_key = key.replace("conv.weight", "conv.conv.weight")
            else:
                raise ValueError(key)
            assert new_key in own, "%s" % new_key
            own[new_key] = value
        self.load_state_dict(own)

    # ------------------------------------------------------------- active sub-network control
    def set_active_subnet(self, wid=None, ks=None, e=None, d=None, pixel_d=None):
        n_mb = len(self.blocks) - len(self.block_group_info[_N_MB_STAGES])
        ks = int2list(ks, n_mb)
        expand = int2list(e, n_mb)
        depth = int2list(d, _N_MB_STAGES)
        pixel_depth = int2list(pixel_d, 1)

        if self.COMPAT_REFERENCE_INDEXING:
            depth.insert(-1, pixel_depth[0])  # Q2: lands on the 4th MB stage; mutates a caller-owned list
        else:
            depth = list(depth) + [pixel_depth[0]]

        for block, k, ratio in zip(self.blocks[:n_mb], ks, expand):
            if k is not None:
                block.mobile_inverted_conv.active_kernel_size = k
            if ratio is not None:
                block.mobile_inverted_conv.active_expand_ratio = ratio
        for i, dd in enumerate(depth):
            if dd is not None:
                self.runtime_depth[i] = min(len(self.block_group_info[i]), dd)

    _CONSTRAINT_SLOTS = {
        "depth": "_depth_include_list", "expand_ratio": "_expand_include_list",
        "kernel_size": "_ks_include_list", "width_mult": "_width_mult_include_list",
        "pixelshuffle_depth": "_pixelshuffle_depth_include_list",
    }

    def set_constraint(self, include_list, constraint_type="depth"):
        if constraint_type not in self._CONSTRAINT_SLOTS:
            raise NotImplementedError
        self.__dict__[self._CONSTRAINT_SLOTS[constraint_type]] = include_list.copy()

    def clear_constraint(self):
        for slot in self._CONSTRAINT_SLOTS.values():
            self.__dict__[slot] = None

    def _candidates(self, slot, default):
        got = self.__dict__.get(slot, None)
        return default if got is None else got

    def sample_active_subnet(self):
        """draws, in this order, one kernel size per MB block, one expand ratio per MB block, one depth
        per MB stage and one pixel-shuffle depth from Python's global `random` -- the exact call
        sequence of the reference (:330-360), so a shared `random.seed(...)` reproduces its samples."""
        n_mb = len(self.blocks) - len(self.block_group_info[_N_MB_STAGES])

        def draw(cands, count):
            if not isinstance(cands[0], list):
                cands = [cands for _ in range(count)]
            return [random.choice(c) for c in cands]

        ks_setting = draw(self._candidates("_ks_include_list", self.ks_list), n_mb)
        expand_setting = draw(self._candidates("_expand_include_list", self.expand_ratio_list), n_mb)
        depth_setting = draw(self._candidates("_depth_include_list", self.depth_list), _N_MB_STAGES)
        pd_setting = draw(self._candidates("_pixelshuffle_depth_include_list", self.pixelshuffle_depth_list), 1)

        self.set_active_subnet(None, ks_setting, expand_setting, depth_setting, pd_setting)
        return {"wid": None, "ks": ks_setting, "e": expand_setting, "d": depth_setting, "pixel_d": pd_setting}

    def re_organize_middle_weights(self, expand_ratio_stage=0):
        """sort every MB block's middle channels by importance (reference ofa_mbs4.py:462-464).  As committed the
        reference walks `self.blocks[2:-2]` -- a slice inherited from the classification supernet, whose first block
        is a fixed stem -- so the FIRST TWO MB blocks of the SR net are never re-organised (pinned by
        tests/golden/reorganize.npz); COMPAT_REFERENCE_INDEXING=False re-organises all of them."""
        n_mb = len(self.blocks) - len(self.block_group_info[_N_MB_STAGES])
        first = 2 if self.COMPAT_REFERENCE_INDEXING else 0
        for block in self.blocks[first:n_mb]:
            block.mobile_inverted_conv.re_organize_middle_weights(expand_ratio_stage)
