"""cProfile of the host side of the training step at a tiny problem size (the GPU is never the limit there)."""
import cProfile, importlib, io, os, pstats, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
amd = lambda m: importlib.import_module("ofa-for-super-resolution_amd." + m)
dop = amd("elastic_nn.modules.dynamic_op"); nets = amd("elastic_nn.networks")
dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
dev = "cuda:0"
net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[6], depth_list=[4], pixelshuffle_depth_list=[2]).to(dev).train()
decay = list(net.get_parameters(["bn", "bias"], mode="exclude")); nod = list(net.get_parameters(["bn", "bias"], mode="include"))
opt = torch.optim.Adam([{"params": decay, "weight_decay": 3e-5}, {"params": nod, "weight_decay": 0}], lr=1e-3, fused=True)
hr = torch.rand(1, 3, 64, 64, device=dev); lr = torch.rand(1, 3, 16, 16, device=dev)
def step(i):
    opt.zero_grad(set_to_none=True)
    random.seed(int("%d%.3d%.3d" % (i, 0, 0)))
    net.sample_active_subnet()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(lr)
    loss = F.mse_loss(out.float(), hr)
    loss.backward()
    opt.step()
for i in range(10): step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(10, 60): step(i)
torch.cuda.synchronize()
print("host floor: %.3f ms/step (tiny tensors, 50 steps, no profiler)" % ((time.perf_counter() - t0) * 1e3 / 50))
# backward nodes run on the autograd engine's device thread, which cProfile does not see: single-threaded mode for the profile
torch.autograd.set_multithreading_enabled(False)
tf = tb = to = 0.0
for i in range(60, 110):
    opt.zero_grad(set_to_none=True)
    random.seed(int("%d%.3d%.3d" % (i, 0, 0)))
    net.sample_active_subnet()
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(lr)
    loss = F.mse_loss(out.float(), hr)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; to += t3 - t2
torch.cuda.synchronize()
print("host time per step: forward %.3f ms, backward %.3f ms, optimizer (+ deferred flush) %.3f ms" % (tf * 20, tb * 20, to * 20))
pr = cProfile.Profile(); pr.enable()
for i in range(10, 40): step(i)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:9000])
