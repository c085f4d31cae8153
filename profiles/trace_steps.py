#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV over the last N training steps (steps are delimited by the
optimizer's multi_tensor_apply launches), optionally skipping the last `skip` steps (bench.py ends with 7 instrumented
per-op steps for the roofline figure).  usage: trace_steps.py <kernel_trace.csv> [nsteps] [top] [skip] [summary.json key]
With the last two arguments the per-kernel averages are also merged into summary.json under `key` (e.g. "bf16", "f32"),
keyed by the kernel symbol as bench.py's kernel table spells it (no "void ofasr::", no parameter list): this is the
file bench.py's roofline.trace_avg_us is read from."""
import collections
import csv
import json
import os
import sys


def site_name(n):
    """rocprofv3's kernel name -> the launch-site name of the library (csrc/api.hip register_site)"""
    depth, cut = 0, None
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    if cut is not None:
        n = n[:cut]
    if n.startswith("void "):
        n = n[5:]
    if n.startswith("ofasr::"):
        n = n[7:]
    return n


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    names = [e[2] for e in ev]
    # the optimizer's launches delimit the steps: the fused Adam's multi-tensor kernels when present (other multi-tensor
    # launches -- the batched BatchNorm counter bump in the forward -- must not split a step)
    opt = [i for i, n in enumerate(names) if "multi_tensor_apply" in n and "Adam" in n]
    if not opt:
        opt = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
    ends = [i for j, i in enumerate(opt) if j == len(opt) - 1 or opt[j + 1] - i > 50]
    lo, hi = ends[-(nsteps + 1 + skip)] + 1, ends[-1 - skip] + 1
    sel = ev[lo:hi]
    span = (sel[-1][1] - sel[0][0]) / 1e6
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        agg[n][0] += e - s
        agg[n][1] += 1
    busy = sum(v[0] for v in agg.values()) / 1e6
    # union of the kernel intervals (kernels overlap when the side stream is active): GPU time with >= 1 kernel running
    union, cur_s, cur_e = 0, None, None
    for s0, e0, _ in sel:
        if cur_e is None or s0 > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    union += (cur_e - cur_s) if cur_e is not None else 0
    print("steps=%d  wall/step=%.3f ms  gpu-busy/step=%.3f ms (sum of kernel durations)  gpu-active/step=%.3f ms "
          "(union)  kernels/step=%.0f" % (nsteps, span / nsteps, busy / nsteps, union / 1e6 / nsteps, len(sel) / nsteps))
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%-96s n/step=%6.1f ms/step=%7.3f avg_us=%8.1f" % (n[:96], c / nsteps, t / 1e6 / nsteps, t / 1e3 / c))
    if len(sys.argv) > 6:
        out, key = sys.argv[5], sys.argv[6]
        table = {}
        if os.path.exists(out):
            table = json.load(open(out))
        ser = key.endswith("_serialized")
        table[key] = {"source": ("rocprofv3 --kernel-trace --pmc FETCH_SIZE of `bench.py --dtype %s` (dispatches serialized: each "
                                 "kernel alone on the GPU), last %d timed steps" % (key[:-len("_serialized")], nsteps)) if ser else
                                "rocprofv3 --kernel-trace of `bench.py --dtype %s`, last %d timed steps" % (key, nsteps),
                      "wall_ms_per_step": round(span / nsteps, 3), "gpu_active_ms_per_step": round(union / 1e6 / nsteps, 3),
                      "kernels": {site_name(n): {"avg_us": round(t / 1e3 / c, 2), "launches_per_step": round(c / nsteps, 2),
                                                 "ms_per_step": round(t / 1e6 / nsteps, 4)}
                                  for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])}}
        json.dump(table, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
