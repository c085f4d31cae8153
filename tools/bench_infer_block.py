#!/usr/bin/env python3
"""Eval-mode MB block: the one-kernel fused path (ofasr_mbconv_infer) against the composite eval path (un-fused kernels,
ofasr_mbconv_fwd), per (mid, K) at N=16, 64x64 (and --S).  Events on the launch stream around `reps` back-to-back block
calls; reports us per block, algorithmic GB/s (x read once + out written once + shortcut read), the matrix-core TFLOP/s of
the useful 1x1 work and its fraction of the bf16 peak.  usage: python tools/bench_infer_block.py [--reps 30]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--N", type=int, default=16)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    import torch
    ops = importlib.import_module(PKG + ".ops")
    dop = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    dl = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_layers")
    blk = importlib.import_module(PKG + ".imagenet_codebase.networks")
    layers = importlib.import_module(PKG + ".layers")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    dev = "cuda:0"
    layer = dl.DynamicMBConvLayer([64], [64], [3, 5, 7], [3, 4, 6])
    block = blk.MobileInvertedResidualBlock(layer, layers.IdentityLayer([64], [64])).to(dev).eval()
    x = torch.randn(a.N, 64, a.S, a.S, device=dev).to(dt)
    px = a.N * a.S * a.S
    for e in (6, 4, 3):
        for K in (7, 5, 3):
            layer.active_kernel_size, layer.active_expand_ratio = K, e
            mid = layer.active_middle_channel(64)
            C = importlib.import_module(PKG + "._C")
            row, ktime = [], {}
            for fused in (True, False):
                ops.FUSED_INFER = fused
                with torch.no_grad():
                    for _ in range(3):
                        block(x)
                    torch.cuda.synchronize()
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for _ in range(a.reps):
                        block(x)
                    t1.record()
                    torch.cuda.synchronize()
                row.append(1e3 * t0.elapsed_time(t1) / a.reps)
                # device time of the kernels themselves (events around every launch, include/ofasr.h Diagnostics): the
                # python-level number above is host-bound at these sizes
                C.lib().ofasr_profile_enable(1)
                with torch.no_grad():
                    for _ in range(a.reps):
                        block(x)
                C.lib().ofasr_profile_enable(0)
                prof = C.profile_read()
                ktime[fused] = sum(v["total_us"] for v in prof.values()) / a.reps
                if fused:
                    main = [v for k, v in prof.items() if k.startswith("mb_fused_kernel")][0]
                    kmain = main["total_us"] / main["launches"]
            ops.FUSED_INFER = True
            tfk = 2.0 * px * 2 * 64 * mid / (kmain * 1e-6) / 1e12
            print("   kernels only: fused block %.1f us (mb_fused_kernel %.1f us = %.0f GB/s, 1x1 %.0f TFLOP/s = %.1f %% of peak) | "
                  "composite eval kernels %.1f us | x%.2f" % (ktime[True], kmain, 3.0 * px * 128 / (kmain * 1e-6) / 1e9, tfk,
                                                            100 * tfk / 2500.0, ktime[False], ktime[False] / ktime[True]))
            us = row[0]
            gb = 3.0 * px * 64 * 2 / (us * 1e-6) / 1e9
            tf = 2.0 * px * 2 * 64 * mid / (us * 1e-6) / 1e12
            print("mid %3d k%d: fused %7.1f us (%6.0f GB/s algorithmic, 1x1 %6.1f TFLOP/s = %4.1f %% of 2.5 PF) | composite %7.1f us | x%.2f"
                  % (mid, K, us, gb, tf, 100 * tf / 2500.0, row[1], row[1] / us))


if __name__ == "__main__":
    main()
