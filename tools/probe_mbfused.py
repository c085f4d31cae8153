#!/usr/bin/env python3
"""mb_fused_kernel time per tile shape (OFASR_MBFUSED_TILE is read once per process, so run once per shape):
usage: OFASR_MBFUSED_TILE=16|32|64 python tools/probe_mbfused.py [HxW ...]"""
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
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 64), (48, 48), (120, 128), (125, 90)]
for (H, W) in shapes:
    N = 16
    x = torch.randn(N, 64, H, W, device=dev).bfloat16()
    for K in (3, 5, 7):
        for mid in (32, 384):
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
            m = [(k, v) for k, v in prof.items() if k.startswith("mb_fused_kernel")][0]
            print("tile %s  %3dx%-3d k%d mid %3d: %7.1f us   %s" % (os.environ.get("OFASR_MBFUSED_TILE", "auto"), H, W, K, mid,
                                                                m[1]["total_us"] / m[1]["launches"], m[0][:60]), flush=True)
