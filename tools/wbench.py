#!/usr/bin/env python3
"""Write / copy ceilings of the box for the roofline discussion: fill and copy of 50 MB (one mid tensor in bf16) and
400 MB, timed with events over 50 reps; plus the expand 1x1 (fan-out kernel) with cold inputs/outputs (a 600 MB fill
between launches evicts L2 and the 256 MB infinity cache)."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=50, between=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if between is not None:
            between()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return 1e3 * tot / reps


def main():
    dev = "cuda:0"
    for mb in (50, 400):
        n = mb * 1000 * 1000 // 2
        a = torch.empty(n, dtype=torch.bfloat16, device=dev)
        b = torch.empty(n, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda: a.zero_())
        print("fill  %4d MB: %7.1f us  %6.0f GB/s written" % (mb, us, mb * 1e6 / us / 1e3))
        us = timeit(lambda: b.copy_(a))
        print("copy  %4d MB: %7.1f us  %6.0f GB/s read+written" % (mb, us, 2 * mb * 1e6 / us / 1e3))
    C = importlib.import_module("ofa-for-super-resolution_amd._C")
    L = C.lib()
    N, S, mid = 16, 64, 384
    HW = S * S
    x = torch.randn(N, 64, S, S, device=dev).to(torch.bfloat16)
    y = torch.empty(N, mid, S, S, device=dev, dtype=torch.bfloat16)
    w = torch.randn(mid, 64, device=dev) * 0.1
    big = torch.empty(300 * 1000 * 1000, dtype=torch.bfloat16, device=dev)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fn = lambda: L.ofasr_pwconv_fwd(P(x), P(w), 64, P(y), N, 64, mid, HW, 2, st())
    nb = (x.numel() + y.numel()) * 2
    us = timeit(fn)
    print("expand 1x1 warm: %6.1f us  %6.0f GB/s algorithmic" % (us, nb / us / 1e3))
    us = timeit(fn, reps=20, between=lambda: big.zero_())
    print("expand 1x1 cold: %6.1f us  %6.0f GB/s algorithmic" % (us, nb / us / 1e3))
    x2 = torch.randn(N, mid, S, S, device=dev).to(torch.bfloat16)
    y2 = torch.empty(N, 64, S, S, device=dev, dtype=torch.bfloat16)
    w2 = torch.randn(64, mid, device=dev) * 0.1
    fn2 = lambda: L.ofasr_pwconv_fwd(P(x2), P(w2), mid, P(y2), N, mid, 64, HW, 2, st())
    us = timeit(fn2)
    print("project 1x1 warm: %6.1f us  %6.0f GB/s algorithmic" % (us, nb / us / 1e3))
    us = timeit(fn2, reps=20, between=lambda: big.zero_())
    print("project 1x1 cold: %6.1f us  %6.0f GB/s algorithmic" % (us, nb / us / 1e3))


if __name__ == "__main__":
    main()
