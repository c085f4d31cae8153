#!/usr/bin/env python3
"""Per-stream picture of the timed training steps from a rocprofv3 kernel-trace CSV: for each HIP stream (queue), the
busy time per step and when its last kernel of the step ends relative to the step's last kernel -- shows which of the
main / side streams of the composite backward is the critical one.
usage: stream_balance.py <kernel_trace.csv> [nsteps] [skip]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(csv.DictReader(open(path)))
    qkey = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[qkey]) for r in rows)
    opt = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] and "Adam" in e[2]]
    if not opt:
        opt = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2]]
    ends = [i for j, i in enumerate(opt) if j == len(opt) - 1 or opt[j + 1] - i > 50]
    bounds = ends[-(nsteps + 1 + skip):len(ends) - skip]
    per = collections.defaultdict(lambda: [0.0, 0.0, 0, 0.0])   # busy, tail gap, kernels, first-start offset
    for a, b in zip(bounds[:-1], bounds[1:]):
        sel = ev[a + 1:b + 1]
        t0 = sel[0][0]
        adam = [s for s, e, n, q in sel if "multi_tensor_apply" in n and "Adam" in n]
        first_adam = min(adam) if adam else min(s for s, e, n, q in sel if "multi_tensor_apply" in n)
        last = collections.defaultdict(int)
        first = {}
        for s, e, n, q in sel:
            per[q][0] += e - s
            per[q][2] += 1
            last[q] = max(last[q], e)
            first.setdefault(q, s)
        for q in last:
            per[q][1] += first_adam - last[q]      # > 0: the stream was done this long before the optimizer started
            per[q][3] += first[q] - t0
    print("stream key: %s; per step over %d steps" % (qkey, nsteps))
    for q, (busy, gap, n, f0) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print("stream %-6s kernels/step=%6.1f busy=%7.3f ms  first kernel at +%7.3f ms  idle before optimizer=%7.3f ms" % (
            q, n / nsteps, busy / 1e6 / nsteps, f0 / 1e6 / nsteps, gap / 1e6 / nsteps))


if __name__ == "__main__":
    main()
