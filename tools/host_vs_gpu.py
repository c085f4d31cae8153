"""Is the bench step bound by the host or by the GPU?  Times K steps twice: until the host has ENQUEUED them (no sync) and
until the GPU has finished them.  usage: python tools/host_vs_gpu.py [--config c3] [--dtype bf16] [--steps 30]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
M = bench.mods()
M["ops"].single_thread_backward(True)
wl = bench.TrainWorkload(M, a.config, dev, 16, 48 if a.config == "c2" else 64, a.dtype)
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(a.steps):
        wl.step(8 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms/step   gpu done %.3f ms/step   (gpu lag at the end %.3f ms)" %
          ((t1 - t0) / a.steps * 1e3, (t2 - t0) / a.steps * 1e3, (t2 - t1) * 1e3), flush=True)
