#!/usr/bin/env python3
"""Host-side profile of the config-5 evaluation step (bench.py EvalWorkload): cProfile of 20 steps with the GPU queue
kept short (a synchronize per step), top functions by own and cumulative time, and the step time with / without the
synchronize.  usage: python tools/host_profile_eval.py"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

M = bench.mods()
wl = bench.EvalWorkload(M, "cuda:0", "bf16")
for i in range(5):
    wl.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    wl.step(i)
torch.cuda.synchronize()
print("free-running: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
th = 0.0
for i in range(20):
    a = time.perf_counter()
    wl.step(i)
    th += time.perf_counter() - a
    torch.cuda.synchronize()
print("host only (enqueue time, GPU idle at step start): %.3f ms/step; with sync %.3f" % (th / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    wl.step(i)
    torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:44]))
