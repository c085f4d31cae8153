#!/usr/bin/env python3
"""Sampled sub-network evaluation -- counterpart of the reference's eval_ofa_net_sr.py (:187-220,247-251): build
OFAMobileNetS4(k7,e6,d4,pd2), load a checkpoint, fix a sub-network and report (loss, Y-PSNR) on the test loader
(BASELINE config 5; full-resolution images, sides multiples of 4).  Images of equal size are batched together
(--batched, the default: SRRunManager.validate_batched; per-image loss / PSNR, identical to the batch-1 pass) and the
pass is timed: images/s of the evaluation, after one untimed warm-up pass."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--path", default="exp/sr/eval")
    ap.add_argument("--ks", type=int, default=7)
    ap.add_argument("--expand", type=int, default=6)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--pixelshuffle-depth", type=int, default=2)
    ap.add_argument("--mix-prec", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--synthetic", action="store_true", help="evaluate on synthetic Set14-sized images when the "
                                                             "dataset directory is absent")
    ap.add_argument("--batch1", action="store_true", help="the reference's batch-1 pass instead of size buckets")
    a = ap.parse_args()
    import torch
    rm = importlib.import_module(PKG + ".imagenet_codebase.run_manager")
    nets = importlib.import_module(PKG + ".elastic_nn.networks")
    dop = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                              pixelshuffle_depth_list=[1, 2])
    # Set14-like sizes (HR sides multiples of 4)
    cfg = rm.Div2K_SetXXRunConfig(n_epochs=1, init_lr=1e-3, opt_type="adam", no_decay_keys="bn#bias",
                                  label_smoothing=0.0, train_batch_size=1, test_batch_size=1, image_size=256,
                                  test_sizes=[(480, 500), (576, 720), (512, 512), (288, 352), (360, 248), (276, 276), (360, 500), (288, 352),
                                              (512, 512), (512, 512), (512, 768), (512, 512), (656, 528), (388, 584)],
                                  n_train_batches=1, allow_synthetic=True if a.synthetic else None)
    mgr = rm.SRRunManager(a.path, net, cfg, init=a.checkpoint is None, mix_prec=a.mix_prec, num_gpus=1)
    if a.checkpoint:
        net.load_weights_from_net(torch.load(a.checkpoint, map_location="cpu", weights_only=True)["state_dict"])
    net.set_active_subnet(ks=a.ks, e=a.expand, d=a.depth, pixel_d=a.pixelshuffle_depth)
    key = "4x_down_image" if a.pixelshuffle_depth == 2 or net.COMPAT_REFERENCE_INDEXING else "2x_down_image"
    import time
    n_img = sum(b["image"].shape[0] for b in cfg.test_loader)

    def run():
        if a.batch1:
            return mgr.validate(is_test=True, input_key=key) + (n_img,)
        return mgr.validate_batched(is_test=True, input_key=key)

    run()                                   # warm-up (allocator, MIOpen find for the vendor-path convs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, psnr, calls = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("loss %.5f  Y-PSNR %.3f dB  (%d images in %d forward calls, %.1f images/s)" % (loss, psnr, n_img, calls,
                                                                                       n_img / dt))


if __name__ == "__main__":
    main()
