#!/usr/bin/env python3
"""bench.py -- OFA-SR progressive-shrinking training throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Metric (BASELINE.json): SR training images/sec, 64x64 -> 256x256 (4x), whole-job aggregate.
One "step" = one pass of the hot path over one synthetic mini-batch per GPU: sample a
sub-network (seed rule of the reference, progressive_shrinking.py:164), forward, MSE loss,
backward, gradient all-reduce (N > 1), Adam step.  Inputs are resident in HBM before the timed
region.  Prints ONE JSON line on rank 0 with `roofline` (dominant HIP kernel, timed with events on
the launch stream) and `cpu_baseline` (the CPU oracle port of the same training step, timed on
the host cores in this run).
"""
import argparse
import importlib
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
MFMA_PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (reference train_batch_size 16)")
    ap.add_argument("--lr-size", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--h2d", action="store_true", help="copy the batch from pinned host memory inside every timed step "
                                                       "(the PCIe-inclusive rate quoted in DESIGN.md; never `value`)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                                                      "the N>1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-images", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def subnet_seed(step, sub=0):
    return int("%d%.3d%.3d" % (step, sub, 0))   # progressive_shrinking.py:164


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(torch.cuda.device_count(), 1)   # rehearsal: ranks may share a GPU
            dist.init_process_group(args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    C = importlib.import_module(PKG + "._C")
    C.lib()   # fail loudly if the HIP extension is missing
    ops = importlib.import_module(PKG + ".ops")
    dop = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    nets = importlib.import_module(PKG + ".elastic_nn.networks")
    dd = importlib.import_module(PKG + ".distributed")

    torch.manual_seed(0)
    random.seed(0)
    torch.backends.cudnn.benchmark = True                           # sr_run_manager.py:153 (MIOpen find mode)
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1            # train_ofa_net_sr_simple.py:183
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[6], depth_list=[4], pixelshuffle_depth_list=[2])
    net.init_model("he_fout")
    net.to(dev).train()
    dd.broadcast_module(net)
    n_params = sum(p.numel() for p in net.parameters())

    # optimizer: Adam, weight decay 3e-5 except on names with 'bn' / 'bias' (sr_run_manager.py:180-191)
    decay = list(net.get_parameters(["bn", "bias"], mode="exclude"))
    no_decay = list(net.get_parameters(["bn", "bias"], mode="include"))
    opt = torch.optim.Adam([{"params": decay, "weight_decay": 3e-5}, {"params": no_decay, "weight_decay": 0}],
                           lr=1e-3, fused=True if os.environ.get("OFASR_FUSED_ADAM", "1") != "0" else None)
    reducer = dd.FlatGradReducer(net.parameters(), gather=True) if world > 1 else None

    act_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    N, S = args.batch, args.lr_size
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    hr = torch.rand((N, 3, 4 * S, 4 * S), generator=g)
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True).clamp_(0, 1)
    hr_host, lr_host = (hr.pin_memory(), lr.pin_memory()) if args.h2d else (None, None)
    hr, lr = hr.to(dev), lr.to(dev)

    def train_step(step):
        if args.h2d:   # what the reference's loader hands over: host tensors
            hr.copy_(hr_host, non_blocking=True)
            lr.copy_(lr_host, non_blocking=True)
        if reducer is not None:
            reducer.prepare()
        else:
            opt.zero_grad(set_to_none=True)
        random.seed(subnet_seed(step))
        net.sample_active_subnet()
        if act_dtype == torch.float32:
            out = net(lr)
            loss = F.mse_loss(out, hr)
        else:
            with torch.autocast("cuda", dtype=act_dtype):
                out = net(lr)
            loss = F.mse_loss(out.float(), hr)
        loss.backward()
        if reducer is not None:
            reducer.reduce()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train_step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = world * N * args.steps / dt
    final_loss = float(loss.detach())

    # ---- roofline of the dominant HIP kernel: events on the launch stream around every library launch
    roofline = None
    kernel_table = None
    if not args.no_roofline:   # every rank runs the instrumented steps (they contain the gradient all-reduce); rank 0 reports
        # per-kernel timing needs one API call per kernel: the instrumented steps run the per-op Functions
        # (same kernels, same order) instead of the composite MB-block call used in the timed region
        ops.FUSED_BLOCK = False
        train_step(args.warmup + args.steps)
        ops.TIMER = ops.KernelTimer()
        nprof = min(args.steps, 6)
        for i in range(nprof):
            train_step(args.warmup + args.steps + 1 + i)
        summ = ops.TIMER.summary()
        ops.TIMER = None
        ops.FUSED_BLOCK = True
        kernel_table = {k: {"avg_us": round(v["avg_us"], 2), "launches_per_step": v["launches"] / nprof,
                            "ms_per_step": round(v["total_ms"] / nprof, 4),
                            "GBps": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) if v["total_ms"] > 0 else None}
                        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        # group API calls that are ONE kernel launch by the HIP kernel they run; the dominant one is reported
        groups = {}
        for k, v in summ.items():
            sym = kernel_symbol(k)
            if sym is None:
                continue
            g = groups.setdefault(sym, {"launches": 0, "total_ms": 0.0, "bytes": 0.0, "flops": 0.0, "calls": []})
            for f in ("launches", "total_ms", "bytes", "flops"):
                g[f] += v[f]
            g["calls"].append(k)
        name, top = max(groups.items(), key=lambda kv: kv[1]["total_ms"])
        achieved = top["bytes"] / (top["total_ms"] * 1e-3) / 1e9
        per_launch = top["bytes"] / top["launches"]
        roofline = {"kernel": name, "api_calls": sorted(top["calls"]), "bound": "hbm", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": pmc_traffic(name), "avg_launch_us": round(1e3 * top["total_ms"] / top["launches"], 2),
                    "launches_per_step": top["launches"] / nprof,
                    "algorithmic_bytes_per_launch": per_launch,
                    "mfma_tflops": round(top["flops"] / (top["total_ms"] * 1e-3) / 1e12, 2),
                    "hip_library_ms_per_step": round(sum(v["total_ms"] for v in summ.values()) / nprof, 3)}

    # ---- CPU baseline: the oracle port of the same training step on the host cores (bounded sample)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(args, S)

    if rank == 0:
        line = {
            "metric": "sr_training_images_per_sec_4x_64to256", "value": round(value, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "OFA-SR S4 supernet 4x progressive shrinking (elastic kernel {3,5,7}, e=6, d=4, "
                                   "pd=2), LR %dx%d -> HR %dx%d, fwd+bwd+Adam, train-mode BN" % (S, S, 4 * S, 4 * S),
                       "per_gpu_batch": N, "global_batch": N * world, "params": n_params,
                       "kernel_transform_mode": 1, "compat_reference_indexing": True,
                       "parallelism": "dp%d" % world, "grad_allreduce_bytes": n_params * 4 if world > 1 else 0,
                       "final_loss": final_loss},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels": kernel_table,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def kernel_symbol(call):
    """HIP kernel run by a single-launch API call of the timer table (None for multi-kernel calls)."""
    if call.startswith("pwconv_fwd_") or call.startswith("pwconv_dgrad_"):
        k_red, m_out = (int(v) for v in call.split("_")[2].split("to"))      # reduction width, output rows
        if k_red <= 64:   # 16-bit aligned path: whole 128-row slabs of a 64-wide reduction take the slab-walk kernel
            return "pw_fanout_slabs_kernel" if (k_red == 64 and m_out % 128 == 0 and m_out > 128) else "pw_fanout_kernel"
        return "pw_fanin_pipe_kernel"
    if call.startswith("dwconv_fwd_k") or call.startswith("dwconv_dgrad_k"):
        return "dw_vec_kernel<K=%s>" % call.rsplit("k", 1)[1]
    return {"pixel_shuffle": "ps_r2_kernel", "pixel_unshuffle": "ps_r2_kernel"}.get(call)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes) -- produced by tools/pmc_traffic.py into profiles/pmc_traffic.json."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None
    base = kernel.split("<")[0]
    hit = table.get(kernel) or table.get(base)
    return None if hit is None else hit.get("hbm_bytes_per_launch")


def run_cpu_baseline(args, S):
    """time the CPU oracle port (oracle/s4_port.py, kind "port") of the same training step: same net,
    same sub-network seeds, fp32, Adam -- on `cores` host threads, `cpu_images` images per step."""
    import torch
    import torch.nn.functional as F
    from oracle import s4_port

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))   # a 1-GPU box grants this job a 16-core share (more threads only thrash)
    torch.set_num_threads(cores)
    sd = s4_port.he_fout_state_dict(seed=0)
    params = []
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.Adam(params, lr=1e-3)
    arch = s4_port.Arch(ks_list=(3, 5, 7), expand_list=(6,), depth_list=(4,), pd_list=(2,))
    n = args.cpu_images
    g = torch.Generator().manual_seed(99)
    hr = torch.rand((n, 3, 4 * S, 4 * S), generator=g)
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True).clamp_(0, 1)

    def step(i):
        opt.zero_grad(set_to_none=True)
        random.seed(subnet_seed(i))
        arch.sample_active_subnet()
        out = s4_port.s4_forward(sd, lr, arch, training=True)
        F.mse_loss(out, hr).backward()
        opt.step()

    step(0)
    t0 = time.perf_counter()
    for i in range(args.cpu_steps):
        step(1 + i)
    dt = time.perf_counter() - t0
    return {"value": round(n * args.cpu_steps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d steps x %d images (same net/shapes/seeds as the GPU step, fp32, torch-CPU oracle port)"
                      % (args.cpu_steps, n)}


if __name__ == "__main__":
    main()
