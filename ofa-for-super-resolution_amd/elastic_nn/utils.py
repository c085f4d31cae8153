"""BN helpers used by DynamicMBConvLayer.get_active_subnet / re_organize_middle_weights
(reference ofa/elastic_nn/utils.py:67-82) and BN re-calibration for sampled sub-networks (:16-64)."""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


def adjust_bn_according_to_idx(bn, idx):
    """permute a BatchNorm's channels (reference :67-71)"""
    from .. import ops
    ops.clear_infer_cache()   # .data writes are invisible to the version counters the operand cache keys on
    bn.weight.data = torch.index_select(bn.weight.data, 0, idx)
    bn.bias.data = torch.index_select(bn.bias.data, 0, idx)
    bn.running_mean.data = torch.index_select(bn.running_mean.data, 0, idx)
    bn.running_var.data = torch.index_select(bn.running_var.data, 0, idx)


def copy_bn(target_bn, src_bn):
    """copy the first target.num_features channels (reference :74-82)"""
    from .. import ops
    ops.clear_infer_cache()   # .data writes are invisible to the version counters the operand cache keys on
    n = target_bn.num_features
    target_bn.weight.data.copy_(src_bn.weight.data[:n])
    target_bn.bias.data.copy_(src_bn.bias.data[:n])
    target_bn.running_mean.data.copy_(src_bn.running_mean.data[:n])
    target_bn.running_var.data.copy_(src_bn.running_var.data[:n])


def set_running_statistics(model, data_loader, input_key="image", device=None, distributed=False):
    """Re-estimate BN running statistics of the ACTIVE sub-network (reference :16-64): every
    BatchNorm2d temporarily records the batch mean / variance it sees over `data_loader`, and
    the batch-size-weighted averages (reference AverageMeter.update(val, x.size(0))) of the first
    `feature_dim` channels replace the running statistics.  The reference reads `batch['image']`
    (:55); `input_key` selects another tensor of the SR loaders' dicts.  `distributed` (Horovod
    DistributedTensor in the reference) is accepted for signature parity; BN statistics stay rank-local
    here like everywhere else in the data-parallel path."""
    from .modules.dynamic_op import DynamicBatchNorm2d

    stats = {}
    forward_model = copy.deepcopy(model)
    device = device or next(model.parameters()).device

    def hook(bn, name):
        stats[name] = {"mean_sum": None, "var_sum": None, "n": 0}

        def fwd(x):
            mean = x.mean(dim=(0, 2, 3), keepdim=True)
            var = ((x - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
            rec = stats[name]
            nb = x.size(0)
            m, v = mean.detach().flatten().float() * nb, var.detach().flatten().float() * nb
            rec["mean_sum"] = m if rec["mean_sum"] is None else rec["mean_sum"] + m
            rec["var_sum"] = v if rec["var_sum"] is None else rec["var_sum"] + v
            rec["n"] += nb
            c = mean.size(1)
            return F.batch_norm(x, mean.flatten(), var.flatten(), bn.weight[:c], bn.bias[:c], False, 0.0, bn.eps)

        return fwd

    for name, m in forward_model.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            m.forward = hook(m, name)

    from .. import ops
    from ..utils import device_batch
    flag, fused, fused_infer = DynamicBatchNorm2d.SET_RUNNING_STATISTICS, ops.FUSED_BN, ops.FUSED_INFER
    DynamicBatchNorm2d.SET_RUNNING_STATISTICS = True
    # every BatchNorm2d must go through its (hooked) module forward, static ConvLayers included: neither the fused
    # BatchNorm passes nor the one-kernel inference layers / blocks (which fold the OLD running statistics) may run
    ops.FUSED_BN = False
    ops.FUSED_INFER = False
    try:
        with torch.no_grad():
            for batch in data_loader:
                if isinstance(batch, dict):
                    if input_key not in batch:      # lr_on_device providers hand over {'image_u8'} only
                        batch = device_batch(batch, device)
                    x = batch[input_key]
                else:
                    x = batch[0]
                forward_model(x.to(device))
    finally:
        DynamicBatchNorm2d.SET_RUNNING_STATISTICS = flag
        ops.FUSED_BN = fused
        ops.FUSED_INFER = fused_infer

    for name, m in model.named_modules():
        rec = stats.get(name)
        if rec is None or rec["n"] == 0 or not isinstance(m, nn.BatchNorm2d):
            continue
        c = rec["mean_sum"].numel()
        m.running_mean.data[:c].copy_(rec["mean_sum"] / rec["n"])
        m.running_var.data[:c].copy_(rec["var_sum"] / rec["n"])
    # the .data writes above are invisible to the version counters the inference operand cache (BN-folded weight
    # images) and GraphedEval key on: drop both, or the next eval forward replays the OLD running statistics
    ops.clear_infer_cache()
