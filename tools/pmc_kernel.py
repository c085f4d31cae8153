#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory.  usage: pmc_kernel.py <dir> <kernel substring>"""
import collections, csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    t = v[len(v) // 2:]
    print("%-60s %-28s %14.1f  (n=%d)" % (k, c, sum(t) / len(t), len(v)))
