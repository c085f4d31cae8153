#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md prescribes) into per-kernel HBM bytes per launch.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- python3 bench.py ...
    python tools/pmc_traffic.py out/fetch out/write profiles/pmc_traffic.json [--merge]

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) -> doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.  Only launches from the steady state (the second half
of each kernel's dispatches) are averaged.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def counter_rows(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, counter):
    acc = collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        tail = v[len(v) // 2:]
        out[k] = (sum(tail) / len(tail), len(v))
    return out


def short(name):
    """the key bench.py looks a kernel up by: the demangled symbol without 'void ', the leading 'ofasr::' and the
    parameter list -- e.g. 'bn_bwd_reduce_kernel<ofasr::bf16_t, true, 1, false>' (what ofasr_profile_read reports)"""
    n = name.strip()
    if n.startswith("void "):
        n = n[5:]
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
    if n.startswith("ofasr::"):
        n = n[7:]
    return n


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch = per_kernel(counter_rows(fetch_dir), "FETCH_SIZE")
    write = per_kernel(counter_rows(write_dir), "WRITE_SIZE")
    table = {}
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k in set(fetch) | set(write):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        n = max(nf, nw)
        a = agg[short(k)]
        a[0] += f * n
        a[1] += w * n
        a[2] += n
    for k, (f, w, n) in agg.items():
        if n == 0:
            continue
        fetch_b = 2.0 * 1024.0 * f / n      # KiB -> B, x2: FETCH_SIZE under-counts wide streaming reads on gfx950
        write_b = 1024.0 * w / n
        table[k] = {"launches": n, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
                    "hbm_bytes_per_launch": fetch_b + write_b}
    if len(sys.argv) > 4 and sys.argv[4] == "--merge" and os.path.exists(out):
        # a second command's passes (the fp32 step, config 5) added to the table of the first: kernels the first
        # command already measured keep its numbers
        old = json.load(open(out))
        table = dict(list(table.items()) + list(old.items()))
    with open(out, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]:
        print("%-40s launches=%6d  fetch=%8.2f MB  write=%8.2f MB" % (k, v["launches"], v["fetch_bytes_per_launch"] / 1e6,
                                                                       v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
