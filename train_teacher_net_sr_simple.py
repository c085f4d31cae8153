#!/usr/bin/env python3
"""Fixed-architecture ("teacher") SR training -- counterpart of the reference's
train_teacher_net_sr_simple.py (:79-126,186-242): OFAMobileNetS4(ks=[5], e=[3], d=[2], pd=[1]), frozen-BN epochs
via SRRunManager.train (BASELINE config 1 is this loop on 32x32 crops, batch 4, on the CPU reference)."""
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
    ap.add_argument("--path", default="exp/sr/teacher")
    ap.add_argument("--n-epochs", type=int, default=120)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--image-size", type=int, default=128)
    ap.add_argument("--mix-prec", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--ks", type=int, default=5)
    ap.add_argument("--expand", type=int, default=3)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--pixelshuffle-depth", type=int, default=1)
    ap.add_argument("--synthetic", action="store_true", help="train on synthetic images when DIV2K is absent")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(lr)
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    rm = importlib.import_module(PKG + ".imagenet_codebase.run_manager")
    nets = importlib.import_module(PKG + ".elastic_nn.networks")
    args = argparse.Namespace(teacher_model=None, kd_ratio=0, kd_type="ce")
    cfg = rm.Div2K_SetXXRunConfig(n_epochs=a.n_epochs, init_lr=1e-3, opt_type="adam", weight_decay=3e-5,
                                  no_decay_keys="bn#bias", label_smoothing=0.0, train_batch_size=a.batch,
                                  test_batch_size=1, image_size=a.image_size, n_worker=8,
                                  allow_synthetic=True if a.synthetic else None)
    net = nets.OFAMobileNetS4(ks_list=[a.ks], expand_ratio_list=[a.expand], depth_list=[a.depth],
                              pixelshuffle_depth_list=[a.pixelshuffle_depth])
    mgr = rm.SRRunManager(a.path, net, cfg, mix_prec=a.mix_prec, num_gpus=int(os.environ.get("WORLD_SIZE", "1")),
                          args=args)
    mgr.save_config()
    mgr.train(args)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
