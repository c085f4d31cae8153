#!/usr/bin/env python3
"""bench.py -- OFA-SR supernet hot path on MI355X: training throughput (BASELINE metric) and the other BASELINE configs.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

--config c3 (default; BASELINE.json `metric`): SR training images/sec, 64x64 -> 256x256 (4x), whole-job aggregate.
    One "step" = one pass of the hot path over one synthetic mini-batch per GPU: sample a sub-network (seed rule of the
    reference, progressive_shrinking.py:164), forward, MSE loss, backward, gradient all-reduce (N > 1), Adam step.
    Inputs are resident in HBM before the timed region.
--config c2: S4 2x supernet, fixed sub-network k=3 / d=4 / e=6, LR 48x48 -> HR 96x96, one training step per step.
--config c5: sampled sub-network (ks=7, e=6, d=2, pixel_d=2) 4x inference over the 14 Set14 image sizes, eval mode,
    images of equal size batched together (eval_ofa_net_sr.py); a "step" is one pass over the 14 images.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the kernel of the HIP library with the largest total time (ALL library kernels compete, on whichever stream
                they ran; ranked by the committed SERIALIZED trace of this command -- the concurrent ranking flips between
                runs -- with the next two as `runners_up`), timed live with HIP events attached to each kernel's own dispatch
                (hipExtLaunchKernel start / stop events, ofasr_profile_enable in include/ofasr.h: the kernel's begin -> end
                time as rocprofv3's kernel trace reports it, no barrier packets added, streams overlap as in the timed
                region) on the SAME composite path the timed region runs; `trace_avg_us` beside it is the average of the
                same symbol in the committed rocprofv3 trace of the same command (profiles/trace_summary.json, static),
                `serialized_avg_us` the same with the dispatches serialized (the counter pass: the kernel alone on the
                GPU) -- a kernel's duration in the step depends on what the other stream runs beside it: the serialized
                figure is the kernel's own, the concurrent trace what the step pays, the live events lie between (or
                at the serialized figure for a side-stream kernel that the instrumented steps happen to run alone);
  kernels       the per-kernel table of those profiled steps (symbol = the name rocprofv3 prints);
  pointwise     the 1x1 path's MFMA TFLOP/s against the MFMA peak of the dtype (north_star quotes its target against it);
  fp32          (c3, dtype != f32) the same training step with fp32 activations -- the reference's arithmetic -- as a
                complete leg of its own: value, ms_per_step, roofline, pointwise, kernels (`--dtype f32` makes it the
                headline of the line instead);
  cpu_baseline  the CPU oracle port of the same training step timed on the host cores in this run.
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

# HR sizes (H, W) of the 14 Set14 images after ModCrop(4) (div2k_setxx.py:182-190); LR = HR / 4
SET14_HR = [(480, 500), (576, 720), (512, 512), (288, 352), (360, 248), (276, 276), (360, 500), (288, 352),
            (512, 512), (512, 512), (512, 768), (512, 512), (656, 528), (388, 584)]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="c3", choices=["c3", "c2", "c5"])
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (reference train_batch_size 16)")
    ap.add_argument("--lr-size", type=int, default=None, help="LR crop side (c3: 64, c2: 48)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-activation leg of the c3 line")
    ap.add_argument("--fp32-steps", type=int, default=10)
    ap.add_argument("--h2d", action="store_true", help="copy the batch from pinned host memory inside every timed step "
                                                       "(the PCIe-inclusive rate quoted in DESIGN.md; never `value`)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only to rehearse "
                                                      "the N>1 code path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-graphs", action="store_true", help="c5: eager launches instead of one hipGraph replay per bucket")
    ap.add_argument("--graph-step", action="store_true",
                    help="c2 (fixed sub-network): capture the whole training step into ONE hipGraph and replay it.  Bit-identical "
                         "to eager steps (tests/test_bench_step.py) but SLOWER on this stack -- 4.13 against 3.62 ms per step: the "
                         "replayed graph loses part of the two-stream overlap -- so it is not the default")
    ap.add_argument("--cpu-images", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args(argv)


def subnet_seed(step, sub=0):
    return int("%d%.3d%.3d" % (step, sub, 0))   # progressive_shrinking.py:164


def mods():
    m = {k: importlib.import_module(PKG + "." + k) for k in ("_C", "ops", "distributed")}
    m["dop"] = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    m["nets"] = importlib.import_module(PKG + ".elastic_nn.networks")
    return m


class TrainWorkload(object):
    """the training step of configs c3 / c2: same objects for the timed leg, the fp32 leg and tests/test_bench_step.py"""

    def __init__(self, M, config, dev, batch, lr_size, dtype, world=1, rank=0, h2d=False, seed=0, graph_step=False):
        import torch
        import torch.nn.functional as F
        self.torch, self.F, self.M = torch, F, M
        self.config, self.dev, self.N, self.S, self.world = config, dev, batch, lr_size, world
        self.act_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
        torch.manual_seed(seed)
        random.seed(seed)
        M["ops"].single_thread_backward(True)     # as SRRunManager does: no hand-off to the engine's worker thread
        M["dop"].DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1            # train_ofa_net_sr_simple.py:183
        if config == "c3":
            net = M["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[6], depth_list=[4],
                                           pixelshuffle_depth_list=[2])
            self.scale, self.fixed = 4, None
        else:   # c2: 2x supernet, fixed k=3 / e=6 / d=4
            net = M["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                           pixelshuffle_depth_list=[1])
            self.scale, self.fixed = 2, dict(ks=3, e=6, d=4, pixel_d=1)
        net.init_model("he_fout")
        net.to(dev).train()
        M["distributed"].broadcast_module(net)
        self.net = net
        self.n_params = sum(p.numel() for p in net.parameters())
        # optimizer: Adam, weight decay 3e-5 except on names with 'bn' / 'bias' (sr_run_manager.py:180-191)
        decay = list(net.get_parameters(["bn", "bias"], mode="exclude"))
        no_decay = list(net.get_parameters(["bn", "bias"], mode="include"))
        # c2's sub-network is FIXED, so every step launches the same kernels on the same buffers: with graph_step the whole
        # step (forward, loss, backward, side-stream join, Adam) is captured once into a hipGraph and replayed (opt-in:
        # measured slower than the eager step here); c3 samples a new sub-network per step (3^16 kernel-size assignments)
        self.graph = None
        self.want_graph = bool(graph_step) and self.fixed is not None and world == 1 and not h2d
        self.opt = torch.optim.Adam([{"params": decay, "weight_decay": 3e-5}, {"params": no_decay, "weight_decay": 0}],
                                    lr=1e-3, fused=True if os.environ.get("OFASR_FUSED_ADAM", "1") != "0" else None,
                                    capturable=self.want_graph)
        early = net.early_gradient_parameters() if os.environ.get("OFASR_DP_OVERLAP", "0") != "0" else None
        self.reducer = M["distributed"].FlatGradReducer(net.parameters(), gather=True, early_params=early) if world > 1 else None
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        hr = torch.rand((batch, 3, self.scale * lr_size, self.scale * lr_size), generator=g)
        lr = F.interpolate(hr, scale_factor=1.0 / self.scale, mode="bicubic", antialias=True).clamp_(0, 1)
        self.hr_host, self.lr_host = (hr.pin_memory(), lr.pin_memory()) if h2d else (None, None)
        self.hr, self.lr = hr.to(dev), lr.to(dev)
        if self.fixed is not None:
            net.set_active_subnet(**self.fixed)

    def capture(self):
        """warm up eagerly on a side stream, then capture one whole training step; False when the capture fails (the
        workload then stays eager)"""
        torch = self.torch
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for i in range(3):
                self._eager_step(i)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(g):
                self.static_loss = self._eager_step(0)
            self.graph = g
        except Exception as e:   # noqa: BLE001 -- a failed capture must not take the benchmark down
            sys.stderr.write("whole-step capture failed (%s): eager steps\n" % (e,))
            self.graph = None
            torch.cuda.synchronize()
        return self.graph is not None

    def step(self, i):
        if self.graph is not None:
            self.graph.replay()
            return self.static_loss
        return self._eager_step(i)

    def _eager_step(self, i):
        torch, F = self.torch, self.F
        if self.hr_host is not None:   # what the reference's loader hands over: host tensors
            self.hr.copy_(self.hr_host, non_blocking=True)
            self.lr.copy_(self.lr_host, non_blocking=True)
        if self.reducer is not None:
            self.reducer.prepare()
        else:
            self.opt.zero_grad(set_to_none=True)
        if self.fixed is None:
            random.seed(subnet_seed(i))
            self.net.sample_active_subnet()
        if self.act_dtype == torch.float32:
            out = self.net(self.lr)
            loss = F.mse_loss(out, self.hr)
        else:
            with torch.autocast("cuda", dtype=self.act_dtype):
                out = self.net(self.lr)
            loss = F.mse_loss(out.float(), self.hr)
        if self.reducer is not None:
            self.reducer.arm()           # (a no-op without OFASR_DP_OVERLAP=1: two-bucket overlapped exchange)
        loss.backward()
        self.M["ops"].flush_deferred()   # the MB blocks' weight gradients: one join of the side stream per backward pass
        if self.reducer is not None:
            self.reducer.reduce()
        self.opt.step()
        return loss


class EvalWorkload(object):
    """config c5: eval_ofa_net_sr.py's sub-network over the Set14 sizes, equal sizes batched (size buckets)"""

    def __init__(self, M, dev, dtype, seed=0, graphs=True):
        import torch
        self.torch, self.M, self.dev = torch, M, dev
        self.act_dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
        torch.manual_seed(seed)
        M["dop"].DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
        net = M["nets"].OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                       pixelshuffle_depth_list=[1, 2])
        net.init_model("he_fout")
        net.to(dev).eval()
        net.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)                     # eval_ofa_net_sr.py:207-220
        self.net = net
        self.n_params = sum(p.numel() for p in net.parameters())
        utils = importlib.import_module(PKG + ".utils")
        g = torch.Generator(device="cpu").manual_seed(4321)
        lrs = [torch.rand((1, 3, h // 4, w // 4), generator=g) for (h, w) in SET14_HR]
        self.buckets = [torch.cat(b).to(dev) for b in utils.bucket_by_size(lrs)]
        self.n_images = len(lrs)
        # the product's eval loop (SRRunManager.validate_batched) replays one hipGraph per size bucket for 16-bit
        # inference (graphed.py; OFASR_EVAL_GRAPHS=0 / --no-graphs: eager launches); fp32 is eager there too
        self.graphed = None
        if graphs and self.act_dtype != torch.float32:
            self.graphed = importlib.import_module(PKG + ".graphed").GraphedEval(net, autocast_dtype=self.act_dtype)

    def step(self, i, eager=False):
        torch = self.torch
        outs = []
        with torch.no_grad():
            if self.graphed is not None and not eager:      # the whole pass (all size buckets) as one replayed graph
                return self.graphed.call_many(self.buckets)
            for lr in self.buckets:
                if self.graphed is not None and not eager:
                    outs.append(self.graphed(lr))
                elif self.act_dtype == torch.float32:
                    outs.append(self.net(lr))
                else:
                    with torch.autocast("cuda", dtype=self.act_dtype):
                        outs.append(self.net(lr))
        return outs


def timed(fn, warmup, steps, fence):
    for i in range(warmup):
        fn(i)
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = fn(warmup + i)
    fence()
    return time.perf_counter() - t0, last


def profile_steps(M, fn, first, nprof):
    """run `nprof` more steps with every library launch bracketed by events on its own stream; returns
    {symbol: {launches, total_us, bytes, flops}} (include/ofasr.h, Diagnostics)."""
    C = M["_C"]
    fn(first)                      # settle allocations made by the first profiled launch sites
    C.lib().ofasr_profile_enable(1)
    try:
        for i in range(nprof):
            fn(first + 1 + i)
    finally:
        C.lib().ofasr_profile_enable(0)
    return C.profile_read()


def roofline_from(summ, nprof, dtype):
    table = {}
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_us"]):
        if v["launches"] <= 0:
            continue
        secs = v["total_us"] * 1e-6
        table[k] = {"avg_us": round(v["total_us"] / v["launches"], 2), "launches_per_step": round(v["launches"] / nprof, 2),
                    "ms_per_step": round(v["total_us"] * 1e-3 / nprof, 4),
                    "GBps": round(v["bytes"] / secs / 1e9, 1) if v["bytes"] > 0 else None,
                    "TFLOPs": round(v["flops"] / secs / 1e12, 1) if v["flops"] > 0 else None}
    cands = {k: v for k, v in summ.items() if v["launches"] > 0 and v["bytes"] > 0}
    if not cands:
        return None, table, None
    # the dominant kernel: the library kernel with the largest total time in the committed rocprofv3 trace of this command
    # with the dispatches SERIALIZED (the counter pass: every kernel's own cost; profiles/trace_summary.json key
    # <dtype>_serialized), so that the line and profiles/ talk about the same kernel.  (The ranking of the CONCURRENT
    # trace is not stable: a kernel's duration there depends on what the other stream happens to run beside it, and three
    # consecutive profile runs of the same code put three different kernels on top.)  Without a serialized trace: the
    # concurrent one; without any: the top of the live event table.
    ranking, chosen_by = None, "live event table (largest total time over the profiled steps)"
    for key, what in ((dtype + "_serialized", "serialized (rocprofv3 --kernel-trace --pmc: each kernel alone)"),
                      (dtype, "concurrent (rocprofv3 --kernel-trace)")):
        tr = trace_table(key)
        names = [k for k, _ in sorted(tr.items(), key=lambda kv: -kv[1].get("ms_per_step", 0.0)) if k in cands]
        if names:
            ranking, chosen_by = names, "largest total time in profiles/trace_summary.json [%s], %s, static" % (key, what)
            break
    if ranking is None:
        ranking = [k for k, _ in sorted(cands.items(), key=lambda kv: -kv[1]["total_us"])]
    peak_tf = MFMA_PEAK_TF[dtype]

    def roof_of(name, full):
        top = cands[name]
        secs = top["total_us"] * 1e-6
        # which roof bounds the kernel: its algorithmic intensity against the ridge (peak flops / peak bytes); the
        # one-kernel MB block (9.3 GFLOP over 25 MB at N=16: 370 flop/B, ridge 312) and the static convs sit on the
        # matrix side
        mfma_bound = top["flops"] > 0 and top["flops"] / top["bytes"] > peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
        work = (top["flops"] if mfma_bound else top["bytes"]) / top["launches"]
        peak = (peak_tf * 1e12) if mfma_bound else (HBM_PEAK_GBS * 1e9)
        trace_us, trace_src = trace_avg(name, dtype)
        ser_us, ser_src = trace_avg(name, dtype + "_serialized")
        if mfma_bound:
            ach = top["flops"] / secs / 1e12
            roof = {"kernel": name, "bound": "mfma", "achieved": round(ach, 1), "peak": peak_tf, "unit": "TFLOP/s",
                    "frac": round(ach / peak_tf, 4)}
        else:
            ach = top["bytes"] / secs / 1e9
            roof = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4)}
        roof.update({"avg_launch_us": round(top["total_us"] / top["launches"], 2),
                     "launches_per_step": round(top["launches"] / nprof, 2),
                     # the same symbol in the committed traces: concurrent = what the step pays with the other stream's
                     # kernels on the same CUs; serialized = the kernel alone on the GPU
                     "trace_avg_us": trace_us, "frac_trace": round(work / (trace_us * 1e-6) / peak, 4) if trace_us else None,
                     "serialized_avg_us": ser_us,
                     "frac_serialized": round(work / (ser_us * 1e-6) / peak, 4) if ser_us else None})
        if full:
            traffic, source = pmc_traffic(name)
            roof.update({"traffic": traffic, "traffic_source": source,
                         "algorithmic_bytes_per_launch": top["bytes"] / top["launches"],
                         "hbm_gbps": round(top["bytes"] / secs / 1e9, 1),
                         "mfma_tflops": round(top["flops"] / secs / 1e12, 2) if top["flops"] > 0 else None,
                         "chosen_by": chosen_by, "trace_source": trace_src, "serialized_source": ser_src,
                         "timing": "start/stop events attached to each kernel dispatch (hipExtLaunchKernel), composite "
                                   "path, %d profiled steps, both streams live" % nprof,
                         "hip_library_ms_per_step": round(sum(v["total_us"] for v in summ.values()) * 1e-3 / nprof, 3)})
        return roof

    roof = roof_of(ranking[0], True)
    roof["runners_up"] = [roof_of(k, False) for k in ranking[1:3]]   # the next two by the same ranking, same three figures
    # the 1x1 path (north_star: >= 70 % of the fp16/bf16 MFMA roofline is quoted against this)
    pw = [v for k, v in summ.items() if k.startswith("pw_") and v["flops"] > 0]
    pointwise = None
    if pw:
        t = sum(v["total_us"] for v in pw) * 1e-6
        fl, by = sum(v["flops"] for v in pw), sum(v["bytes"] for v in pw)
        pointwise = {"kernels": "pw_* (expand / project forward, input and weight gradients)",
                     "mfma_tflops": round(fl / t / 1e12, 1), "frac_mfma": round(fl / t / 1e12 / peak_tf, 4),
                     "hbm_gbps": round(by / t / 1e9, 1), "frac_hbm": round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                     "algorithmic_intensity_flop_per_byte": round(fl / by, 1),
                     "ms_per_step": round(t * 1e3 / nprof, 3)}
    return roof, table, pointwise


def trace_table(dtype):
    try:
        with open(os.path.join(ROOT, "profiles", "trace_summary.json")) as f:
            return json.load(f).get(dtype, {}).get("kernels", {})
    except Exception:
        return {}


def trace_avg(kernel, dtype):
    """average duration (us) of `kernel` in the committed rocprofv3 --kernel-trace of the same command
    (profiles/trace_summary.json, written by tools/prof_round.sh through profiles/trace_steps.py --json) -- a static
    cross-check of the live event timing, NOT measured by this run."""
    path = os.path.join(ROOT, "profiles", "trace_summary.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None, None
    sect = table.get(dtype, {})
    hit = sect.get("kernels", {}).get(kernel)
    if hit is None:
        return None, None
    return hit.get("avg_us"), "profiles/trace_summary.json (%s)" % sect.get("source", "rocprofv3 --kernel-trace")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs,
    corrected as MI355X_MICROARCH.md prescribes) -- tools/pmc_traffic.py writes profiles/pmc_traffic.json, keyed by the
    kernel symbol; a static file refreshed with the profiles, NOT measured by this run."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except Exception:
        return None, None
    hit = table.get(kernel)
    if hit is None:
        return None, None
    return hit.get("hbm_bytes_per_launch"), "profiles/pmc_traffic.json (rocprofv3 --pmc passes of the same command)"


def run_cpu_baseline(args, S):
    """time the CPU oracle port (oracle/s4_port.py, kind "port") of the c3 training step: same net, same sub-network
    seeds, fp32, Adam -- on `cores` host threads, `cpu_images` images per step."""
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
            "sample": "%d steps x %d images (same net/shapes/seeds as the GPU c3 step, fp32, torch-CPU oracle port)"
                      % (args.cpu_steps, n)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

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
    torch.backends.cudnn.benchmark = True                           # sr_run_manager.py:153 (MIOpen find mode)

    M = mods()
    M["_C"].lib()   # fail loudly if the HIP extension is missing

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = args.lr_size or (48 if args.config == "c2" else 64)

    def leg(dtype, warmup, steps, h2d):
        """one timed leg (warm-up, EXACTLY `steps` timed steps between two fences, max over ranks) + its profiled steps"""
        if args.config == "c5":
            wl = EvalWorkload(M, dev, dtype, graphs=not args.no_graphs)
            per_step = wl.n_images
        else:
            wl = TrainWorkload(M, args.config, dev, args.batch, S, dtype, world, rank, h2d,
                               graph_step=args.config == "c2" and args.graph_step)
            per_step = args.batch
            if wl.want_graph:
                wl.capture()
        dt, last = timed(wl.step, warmup, steps, fence)
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out = {"value": round(world * per_step * steps / dt, 2), "unit": "images/s",
               "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup, "dtype": dtype,
               "final_loss": float(last.detach()) if args.config != "c5" else None}
        if not args.no_roofline:   # every rank runs the profiled steps (they contain the gradient all-reduce); rank 0 reports
            nprof = min(steps, 6)
            # (a replayed graph runs no host code, so the per-dispatch events need the eager launches of the same kernels)
            prof_step = (lambda i: wl.step(i, eager=True)) if args.config == "c5" else \
                (wl._eager_step if getattr(wl, "graph", None) is not None else wl.step)
            summ = profile_steps(M, prof_step, warmup + steps, nprof)
            out["roofline"], out["kernels"], out["pointwise"] = roofline_from(summ, nprof, dtype)
        else:
            out["roofline"] = out["kernels"] = out["pointwise"] = None
        extra = {"n_params": wl.n_params, "buckets": len(wl.buckets) if args.config == "c5" else None,
                 "graph_step": args.config != "c5" and getattr(wl, "graph", None) is not None,
                 "graphed": args.config == "c5" and wl.graphed is not None, "per_step": per_step}
        del wl
        torch.cuda.empty_cache()
        return out, extra

    main_leg, info = leg(args.dtype, args.warmup, args.steps, args.h2d)
    value, ms_per_step, final_loss = main_leg["value"], main_leg["ms_per_step"], main_leg["final_loss"]
    roofline, kernel_table, pointwise = main_leg["roofline"], main_leg["kernels"], main_leg["pointwise"]
    per_step = info["per_step"]

    fp32 = None
    if args.config == "c3" and args.dtype != "f32" and not args.no_fp32:
        # the reference computes in fp32: the same step with fp32 activations, a complete leg of its own
        fp32, _ = leg("f32", 3, args.fp32_steps, False)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config == "c3":
        cpu_baseline = run_cpu_baseline(args, S)

    if rank == 0:
        if args.config == "c3":
            metric = "sr_training_images_per_sec_4x_64to256"
            workload = ("OFA-SR S4 supernet 4x progressive shrinking (elastic kernel {3,5,7}, e=6, d=4, pd=2), "
                        "LR %dx%d -> HR %dx%d, fwd+bwd+Adam, train-mode BN" % (S, S, 4 * S, 4 * S))
        elif args.config == "c2":
            metric = "sr_training_images_per_sec_2x_48to96"
            workload = ("OFA-SR S4 supernet 2x, fixed sub-network k=3 / d=4 / e=6, LR %dx%d -> HR %dx%d, fwd+bwd+Adam, "
                        "train-mode BN (BASELINE config 2)" % (S, S, 2 * S, 2 * S))
        else:
            metric = "sr_inference_images_per_sec_4x_set14_sizes"
            workload = ("sampled sub-network (ks=7, e=6, d=2, pixel_d=2) 4x inference over the 14 Set14 image sizes "
                        "(LR 62x90 .. 192x128), eval-mode BN, equal sizes batched: %d forward calls per pass "
                        "(BASELINE config 5)" % info["buckets"])
        cfg = {"workload": workload, "params": info["n_params"] if args.config == "c5" else None,
               "kernel_transform_mode": 1, "compat_reference_indexing": True, "parallelism": "dp%d" % world}
        if args.config != "c5":
            cfg.update({"per_gpu_batch": args.batch, "global_batch": args.batch * world,
                        "grad_allreduce_bytes": 2160422 * 4 if world > 1 else 0, "final_loss": final_loss,
                        "launch": "one hipGraph replay per training step (fixed sub-network)" if info.get("graph_step")
                        else "eager"})
            cfg.pop("params")
        else:
            cfg["images_per_pass"] = per_step
            cfg["launch"] = ("ONE hipGraph replay per pass (all size buckets, graphed.GraphedEval.call_many); kernel table from eager launches"
                             if info["graphed"] else "eager")
        line = {
            "metric": metric, "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic", "config": cfg, "roofline": roofline, "pointwise": pointwise, "fp32": fp32,
            "cpu_baseline": cpu_baseline, "kernels": kernel_table,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
