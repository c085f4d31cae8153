"""Progressive-shrinking training of the SR supernet (mirror of reference
ofa/elastic_nn/training/progressive_shrinking.py:24-461): validate / train_one_epoch / train /
load_models and the three stage drivers (elastic depth, expand ratio, pixel-shuffle depth).

Same call signatures and the same arithmetic: per mini-batch, `dynamic_batch_size` sub-networks are
sampled with the deterministic seed int('%d%.3d%.3d' % (epoch*nBatch+i, sub, 0)) (:164) -- identical
on every rank, so all ranks train the same sub-network -- the 2x or 4x LR input is chosen by the
sampled pixel_d (:177-180), gradients accumulate over the sub-steps and Adam steps once.

MI355X-first differences: tensors go to run_manager.device (not a hard-coded .cuda()); the gradient
exchange is run_manager.step() (one flat RCCL all-reduce); PSNR is accumulated on the device and read
once per epoch; checkpoints paths are arguments (`args.teacher_path`) instead of literals edited in source.
"""
import json
import os
import random
import time

import torch
import torch.nn.functional as F

from ... import ops
from ...imagenet_codebase.run_manager.sr_run_manager import SRRunManager  # noqa: F401
from ...utils import AverageMeter, device_batch, int2list, list_mean, psnr_y_device, subset_mean


def validate(run_manager, epoch=0, is_test=True, image_size_list=None, width_mult_list=None, ks_list=None,
             expand_ratio_list=None, depth_list=None, pixelshuffle_depth_list=None, additional_setting=None):
    """evaluate the Cartesian product of sub-network settings (reference :24-91).
    returns (mean loss, mean PSNR, 'PDx-Wx-Dx-Ex-Kx (psnr), ...' log string)."""
    dynamic_net = run_manager.net
    dynamic_net.eval()
    if width_mult_list is None:
        width_mult_list = list(range(len(dynamic_net.width_mult_list)))
    ks_list = dynamic_net.ks_list if ks_list is None else ks_list
    expand_ratio_list = dynamic_net.expand_ratio_list if expand_ratio_list is None else expand_ratio_list
    depth_list = dynamic_net.depth_list if depth_list is None else depth_list
    if pixelshuffle_depth_list is None:
        pixelshuffle_depth_list = dynamic_net.pixelshuffle_depth_list

    settings = []
    for pixel_d in pixelshuffle_depth_list:
        for w in width_mult_list:
            for d in depth_list:
                for e in expand_ratio_list:
                    for k in ks_list:
                        settings.append([{"pixel_d": pixel_d, "wid": w, "d": d, "e": e, "ks": k},
                                         "PD%s-W%s-D%s-E%s-K%s" % (pixel_d, w, d, e, k)])
    if additional_setting is not None:
        settings += additional_setting

    losses, psnrs, valid_log = [], [], ""
    for setting, name in settings:
        run_manager.write_log("-" * 30 + " Validate %s " % name + "-" * 30, "train", should_print=False)
        dynamic_net.set_active_subnet(**setting)
        run_manager.write_log(dynamic_net.module_str, "train", should_print=False)
        # feed the LR image of the scale the active path really up-samples by (the reference always feeds the 2x image,
        # quirk Q4, and under quirk Q1 pixel_d does not decide the scale)
        up = dynamic_net.active_upscale() if hasattr(dynamic_net, "active_upscale") else None
        kw = {} if up is None else {"input_key": "%dx_down_image" % up}
        loss, psnr = run_manager.validate(epoch=epoch, is_test=is_test, run_str=name, net=dynamic_net, **kw)
        losses.append(loss)
        psnrs.append(psnr)
        valid_log += "%s (%.3f), " % (name, psnr)
    return list_mean(losses), list_mean(psnrs), valid_log


def subnet_seed(epoch, nBatch, i, sub):
    return int("%d%.3d%.3d" % (epoch * nBatch + i, sub, 0))


def train_one_epoch(run_manager, args, epoch, warmup_epochs=0, warmup_lr=0):
    """reference :94-224 (hot loop :152-203).  returns (mean loss, mean PSNR)."""
    dynamic_net = run_manager.net
    dynamic_net.train()
    loader = run_manager.run_config.train_loader
    sampler = getattr(loader, "sampler", None)
    if hasattr(sampler, "set_epoch"):      # a rank-sharding sampler replays the same permutation otherwise
        sampler.set_epoch(epoch)
    nBatch = len(loader)
    dev = run_manager.device
    from ... import distributed as dd
    if getattr(args, "independent_distributed_sampling", False) and dd.world_size() > 1:
        # ranks drawing different sub-networks would disagree on which parameters get grad=None: FlatGradReducer
        # decides that from the local touched set, so Adam state would diverge silently across ranks
        raise ValueError("independent_distributed_sampling is not supported with data parallelism: every rank must "
                         "train the same sub-network (shared seed rule, reference progressive_shrinking.py:164)")
    losses, psnr_meter, data_time = AverageMeter(), AverageMeter(), AverageMeter()
    loss_sum = torch.zeros((), device=dev, dtype=torch.float64)
    psnr_sum = torch.zeros((), device=dev, dtype=torch.float64)
    n_seen = 0
    log = []
    end = time.time()
    for i, mini_batch in enumerate(loader):
        data_time.update(time.time() - end)
        if epoch < warmup_epochs:
            new_lr = run_manager.run_config.warmup_adjust_learning_rate(
                run_manager.optimizer, warmup_epochs * nBatch, nBatch, epoch, i, warmup_lr)
        else:
            new_lr = run_manager.run_config.adjust_learning_rate(run_manager.optimizer, epoch - warmup_epochs, i, nBatch)
        mini_batch = device_batch(mini_batch, dev)     # LR images made on the GPU when the loader ships uint8 HR only
        images, x2, x4 = mini_batch["image"], mini_batch["2x_down_image"], mini_batch["4x_down_image"]

        soft_logits = None
        if args.kd_ratio > 0:
            args.teacher_model.train()
            with torch.no_grad():
                soft_logits = args.teacher_model(x4 if getattr(args, "teacher_scale", 4) == 4 else x2).detach()

        run_manager.zero_grad()
        sub_losses, sub_psnrs = [], []
        subnet_str = ""
        for sub in range(args.dynamic_batch_size):
            if getattr(args, "independent_distributed_sampling", False):
                seed = os.getpid() + time.time()
            else:
                seed = subnet_seed(epoch, nBatch, i, sub)
            random.seed(seed)
            settings = dynamic_net.sample_active_subnet()
            subnet_str += "%d: " % sub + ",".join(
                "%s_%s" % (k, "%.1f" % subset_mean(v, 0) if isinstance(v, list) else v) for k, v in settings.items()
            ) + " || "
            # reference :177-180 picks the input by the sampled pixel_d; under quirk Q1 the S4 net up-samples by
            # 2 ** (active shuffle blocks) whatever pixel_d says, so ask the net (same answer whenever they agree)
            if hasattr(dynamic_net, "active_upscale"):
                lr_img = x2 if dynamic_net.active_upscale() == 2 else x4
            else:
                lr_img = x2 if settings["pixel_d"][0] == 1 else x4
            with run_manager.autocast():
                output = run_manager.net(lr_img)
            output = output.float()
            loss = run_manager.train_criterion(output, images)
            if soft_logits is not None:
                loss = (args.kd_ratio * F.mse_loss(output, soft_logits) + loss) * (2 / (args.kd_ratio + 1))
            sub_losses.append(loss.detach())
            sub_psnrs.append(psnr_y_device(output, images))
            if sub == args.dynamic_batch_size - 1:
                run_manager.last_backward_next()
            loss.backward()
            ops.flush_deferred()   # the MB blocks' weight gradients: joined once per backward pass (ops.py)
        run_manager.step()

        n = images.size(0)
        loss_sum += torch.stack(sub_losses).double().mean() * n
        psnr_sum += torch.stack(sub_psnrs).mean() * n
        n_seen += n
        log.append((new_lr, str(seed), subnet_str))
        end = time.time()
    losses.update(float(loss_sum) / max(n_seen, 1), n_seen)     # the only host syncs of the epoch
    psnr_meter.update(float(psnr_sum) / max(n_seen, 1), n_seen)
    run_manager._last_train_log = log
    return losses.avg, psnr_meter.avg


def train(run_manager, args, validate_func=None):
    """reference :227-254"""
    if validate_func is None:
        validate_func = validate
    for epoch in range(run_manager.start_epoch, run_manager.run_config.n_epochs + args.warmup_epochs):
        train_loss, train_psnr = train_one_epoch(run_manager, args, epoch, args.warmup_epochs, args.warmup_lr)
        if (epoch + 1) % args.validation_frequency == 0:
            val_loss, val_acc, _val_log = validate_func(run_manager, epoch=epoch, is_test=True)
            is_best = val_acc > run_manager.best_acc
            run_manager.best_acc = max(run_manager.best_acc, val_acc)
            val_log = "Valid [{0}/{1}] loss={2:.3f}, top-1={3:.3f} ({4:.3f})".format(
                epoch + 1 - args.warmup_epochs, run_manager.run_config.n_epochs, val_loss, val_acc,
                run_manager.best_acc)
            val_log += ", Train top-1 {top1:.3f}, Train loss {loss:.3f}\t".format(top1=train_psnr, loss=train_loss)
            val_log += _val_log
            run_manager.write_log(val_log, "valid", should_print=False)
            run_manager.save_model({"epoch": epoch, "best_acc": run_manager.best_acc,
                                    "optimizer": run_manager.optimizer.state_dict(),
                                    "state_dict": run_manager.net.state_dict()}, is_best=is_best)


def load_models(run_manager, dynamic_net, model_path=None):
    """warm-start the supernet from a (static or dynamic) checkpoint via the key remap of
    load_weights_from_net (reference :257-263)."""
    init = torch.load(model_path, map_location="cpu", weights_only=True)["state_dict"]
    dynamic_net.load_weights_from_net(init)
    run_manager.write_log("Loaded init from %s" % model_path, "valid")


def _stage_info(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {"stage": 0}


def _run_stages(kind, attr, constraint, key, train_func, run_manager, args, validate_func_dict, reorganize=False):
    """shared body of the three supporting_elastic_* drivers (reference :266-328, :331-396, :399-461):
    warm start -> validate -> for each newly supported value: constrain sampling, train, bump the
    '<kind>.stage' counter, save '<kind>_stage<N>.pth.tar', validate."""
    net = run_manager.net
    stage_info_path = os.path.join(run_manager.path, "%s.stage" % kind)
    stage_info = _stage_info(stage_info_path)
    full_list = getattr(net, attr)
    validate_func_dict[key] = sorted(full_list)
    model_path = getattr(args, "teacher_path", None)
    if model_path:
        load_models(run_manager, net, model_path=model_path)
    if reorganize:
        net.re_organize_middle_weights()
    run_manager.write_log("%.3f\t%.3f\t%s" % validate(run_manager, **validate_func_dict), "valid")

    stage_list = sorted(full_list, reverse=True)
    n_stages = len(stage_list) - 1
    others_fixed = {
        "depth": lambda: len(set(net.ks_list)) == 1 and len(set(net.expand_ratio_list)) == 1,
        "expand": lambda: len(set(net.ks_list)) == 1 and len(set(net.depth_list)) == 1,
        "pixelshuffle_depth": lambda: len(set(net.ks_list)) == 1 and len(set(net.expand_ratio_list)) == 1,
    }[kind]
    for current_stage in range(n_stages - 1, n_stages):
        run_manager.write_log("-" * 30 + "Supporting Elastic %s: %s -> %s" % (
            kind, stage_list[:current_stage + 1], stage_list[:current_stage + 2]) + "-" * 30, "valid")
        supported = stage_list[:current_stage + 2]
        validate_func_dict[key] = supported if others_fixed() else sorted({min(supported), max(supported)})
        net.set_constraint(supported, constraint_type=constraint)
        train_func(run_manager, args,
                   lambda _rm, epoch, is_test: validate(_rm, epoch, is_test, **validate_func_dict))
        stage_info["stage"] += 1
        run_manager.start_epoch = 0
        run_manager.best_acc = 0.0
        if reorganize:
            net.re_organize_middle_weights(expand_ratio_stage=stage_info["stage"])
            from ... import distributed as dd
            dd.broadcast_module(net)   # what DistributedRunManager.broadcast() did (reference :389-390, Q5)
        run_manager.save_model(model_name="%s_stage%d.pth.tar" % (kind, stage_info["stage"]))
        if run_manager.is_root:
            with open(stage_info_path, "w") as f:
                json.dump(stage_info, f, indent=4)
        validate_func_dict[key] = sorted(full_list)
        run_manager.write_log("%.3f\t%.3f\t%s" % validate(run_manager, **validate_func_dict), "valid")


def supporting_elastic_depth(train_func, run_manager, args, validate_func_dict):
    _run_stages("depth", "depth_list", "depth", "depth_list", train_func, run_manager, args, validate_func_dict)


def supporting_elastic_expand(train_func, run_manager, args, validate_func_dict):
    _run_stages("expand", "expand_ratio_list", "expand_ratio", "expand_ratio_list", train_func, run_manager, args,
                validate_func_dict, reorganize=True)


def supporting_elastic_pixelshuffle_depth(train_func, run_manager, args, validate_func_dict):
    _run_stages("pixelshuffle_depth", "pixelshuffle_depth_list", "pixelshuffle_depth", "pixelshuffle_depth_list",
                train_func, run_manager, args, validate_func_dict)
