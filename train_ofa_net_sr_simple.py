#!/usr/bin/env python3
"""Progressive-shrinking training of the OFA-SR supernet on MI355X -- counterpart of the reference's
train_ofa_net_sr_simple.py (same `--task {kernel,depth,expand,pixelshuffle_depth}` / `--phase` interface and
the same `args` attribute names, reference :21-132), one process per GPU:

    python train_ofa_net_sr_simple.py --task kernel                                    # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 train_ofa_net_sr_simple.py --task depth --phase 1

Differences from the reference script: hyper-parameters that it edits in source are flags here
(--net s4|x4, --teacher-path, --path, --mix-prec); with --synthetic the synthetic provider stands in for a missing
DIV2K directory (otherwise a missing dataset is an error); multi-GPU is RCCL data parallelism (one flat all-reduce per step)."""
import argparse
import importlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", type=str, default="depth",
                    choices=["kernel", "depth", "expand", "pixelshuffle_depth"])
    ap.add_argument("--phase", type=int, default=1, choices=[1, 2])
    ap.add_argument("--net", default="s4", choices=["s4", "x4"])
    ap.add_argument("--path", default=None)
    ap.add_argument("--teacher-path", default=None)
    ap.add_argument("--mix-prec", default="bf16", choices=["f32", "bf16", "f16"])
    ap.add_argument("--n-epochs", type=int, default=None)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--n-train-batches", type=int, default=8)
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic images when DIV2K is absent")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))

    # ---- the reference's hard-coded settings per task (reference :28-87)
    table = {
        "kernel": dict(path="exp/sr/normal2kernel", dynamic_batch_size=1, n_epochs=120, base_lr=1e-3,
                       ks_list="3,5,7", expand_list="6", depth_list="4", pixelshuffle_depth_list="2"),
        "depth": dict(path="exp/sr/kernel2kernel_depth/phase%d" % args.phase, dynamic_batch_size=2,
                      n_epochs=25 if args.phase == 1 else 120, base_lr=1e-3 if args.phase == 1 else 2.5e-3,
                      ks_list="3,5,7", expand_list="6", depth_list="3,4" if args.phase == 1 else "2,3,4",
                      pixelshuffle_depth_list="2"),
        "expand": dict(path="exp/sr/kernel_depth2kernel_depth_width/phase%d" % args.phase, dynamic_batch_size=4,
                       n_epochs=25 if args.phase == 1 else 120, base_lr=1e-3 if args.phase == 1 else 2.5e-3,
                       ks_list="3,5,7", expand_list="4,6" if args.phase == 1 else "3,4,6", depth_list="2,3,4",
                       pixelshuffle_depth_list="2"),
        "pixelshuffle_depth": dict(path="exp/sr/full_elastic", dynamic_batch_size=4, n_epochs=120, base_lr=1e-3,
                                   ks_list="3,5,7", expand_list="3,4,6", depth_list="2,3,4",
                                   pixelshuffle_depth_list="1,2"),
    }[args.task]
    user_path, user_epochs = args.path, args.n_epochs
    for k, v in table.items():
        setattr(args, k, v)
    args.n_epochs = user_epochs or table["n_epochs"]
    args.path = user_path or table["path"]
    args.manual_seed = 0
    args.lr_schedule_type = "cosine"
    args.base_batch_size = 16
    args.valid_size = None
    args.opt_type = "adam"
    args.momentum, args.no_nesterov = 0.9, False
    args.weight_decay = 3e-5
    args.label_smoothing = 0.0
    args.no_decay_keys = "bn#bias"
    args.fp16_allreduce = False
    args.model_init = "he_fout"
    args.validation_frequency = 1
    args.print_frequency = 10
    args.n_worker = 8
    args.resize_scale, args.distort_color = 0.35, None      # reference :115 (colour jitter off)
    args.continuous_size, args.not_sync_distributed_image_size = True, False
    args.bn_momentum, args.bn_eps = 0.1, 1e-5
    args.dropout = 0.1
    args.width_mult_list = "1.0"
    args.dy_conv_scaling_mode = 1
    args.independent_distributed_sampling = False
    args.kd_ratio, args.kd_type, args.teacher_model = 0, "ce", None
    args.warmup_epochs, args.warmup_lr = 0, -1
    args.init_lr = args.base_lr              # the SR scripts do not scale LR with the GPU count (reference :168)
    args.train_batch_size = args.base_batch_size
    args.test_batch_size = 1

    torch.manual_seed(args.manual_seed)
    np.random.seed(args.manual_seed)
    random.seed(args.manual_seed)

    rm = importlib.import_module(PKG + ".imagenet_codebase.run_manager")
    dop = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    nets = importlib.import_module(PKG + ".elastic_nn.networks")
    ps = importlib.import_module(PKG + ".elastic_nn.training.progressive_shrinking")

    args.allow_synthetic = True if args.synthetic else None
    run_config = rm.Div2K_SetXXRunConfig(**args.__dict__)
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = args.dy_conv_scaling_mode
    to_list = lambda s, f: [f(x) for x in s.split(",")]
    args.ks_list = to_list(args.ks_list, int)
    args.expand_list = to_list(args.expand_list, int)
    args.depth_list = to_list(args.depth_list, int)
    args.pixelshuffle_depth_list = to_list(args.pixelshuffle_depth_list, int)
    Net = nets.OFAMobileNetS4 if args.net == "s4" else nets.OFAMobileNetX4
    net = Net(bn_param=(args.bn_momentum, args.bn_eps), dropout_rate=args.dropout, width_mult_list=1.0,
              ks_list=args.ks_list, expand_ratio_list=args.expand_list, depth_list=args.depth_list,
              pixelshuffle_depth_list=args.pixelshuffle_depth_list)
    run_manager = rm.SRRunManager(args.path, net, run_config, mix_prec=args.mix_prec,
                                  num_gpus=int(os.environ.get("WORLD_SIZE", "1")), args=args)
    run_manager.save_config()

    validate_func_dict = {"image_size_list": None, "width_mult_list": None,
                          "ks_list": sorted({min(args.ks_list), max(args.ks_list)}),
                          "expand_ratio_list": sorted({min(args.expand_list), max(args.expand_list)}),
                          "depth_list": sorted({min(net.depth_list), max(net.depth_list)}),
                          "pixelshuffle_depth_list": sorted(set(net.pixelshuffle_depth_list))}
    if args.task == "kernel":
        validate_func_dict["ks_list"] = sorted(args.ks_list)
        if args.teacher_path:
            ps.load_models(run_manager, net, args.teacher_path)
        run_manager.write_log("%.3f\t%.3f\t%s" % ps.validate(run_manager, **validate_func_dict), "valid")
        ps.train(run_manager, args, lambda _rm, epoch, is_test: ps.validate(_rm, epoch, is_test, **validate_func_dict))
    elif args.task == "depth":
        ps.supporting_elastic_depth(ps.train, run_manager, args, validate_func_dict)
    elif args.task == "expand":
        ps.supporting_elastic_expand(ps.train, run_manager, args, validate_func_dict)
    else:
        ps.supporting_elastic_pixelshuffle_depth(ps.train, run_manager, args, validate_func_dict)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
