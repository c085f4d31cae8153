#!/usr/bin/env python3
"""mb_fused_kernel time against the number of 32-channel chunks (mid = 32 .. 384) and the image count: separates the
fixed cost (prologue, epilogue) from the per-chunk cost.  usage: python tools/probe_mbfused.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"
import torch  # noqa: E402

ops = importlib.import_module(PKG + ".ops")
C = importlib.import_module(PKG + "._C")
dev = "cuda:0"
for N in (16, 8):
    x = torch.randn(N, 64, 64, 64, device=dev).bfloat16()
    for K in (3, 7):
        for mid in (32, 64, 128, 192, 384):
            bns = [torch.nn.BatchNorm2d(c).to(dev).eval() for c in (mid, mid, 64)]
            w1 = torch.randn(mid, 64, 1, 1, device=dev) * 0.1
            w2 = torch.randn(64, mid, 1, 1, device=dev) * 0.1
            wdw = torch.randn(mid, 1, K, K, device=dev) * 0.1
            cfg = {"mid": mid, "out": 64, "K": K, "chain": [K], "residual": True, "bns": tuple(bns)}
            args = (x, cfg, w1, bns[0].weight, bns[0].bias, wdw, bns[1].weight, bns[1].bias, w2, bns[2].weight, bns[2].bias)
            with torch.no_grad():
                for _ in range(3):
                    ops.mbconv_infer(*args)
                torch.cuda.synchronize()
                C.lib().ofasr_profile_enable(1)
                for _ in range(10):
                    ops.mbconv_infer(*args)
                C.lib().ofasr_profile_enable(0)
            prof = C.profile_read()
            m = [v for k, v in prof.items() if k.startswith("mb_fused_kernel")][0]
            print("N %2d k%d mid %3d (%2d chunks): %7.1f us" % (N, K, mid, mid // 32, m["total_us"] / m["launches"]))
