#!/usr/bin/env python3
"""Static scan of the library's gfx950 ISA for memory requests that are waited for one at a time.

A global load under a lane-dependent bounds test (`if (ok) v = *p;`), or a rolled loop of one load + one LDS store per
iteration, is compiled into  global_load ... s_waitcnt vmcnt(0) ... global_load : every request is a full memory round
trip before the next one is issued.  For each kernel this counts the `s_waitcnt vmcnt(0)` that have a global / buffer
load within 30 instructions on BOTH sides, beside the kernel's total number of loads.  (Static: a flagged path may be one
the hot shapes never take -- check the source; runs on the CPU container, hipcc cross-compiles.)

usage: python tools/scan_serial_loads.py [min_count] [file.hip ...]      default: every csrc/*.hip, min_count 2"""
import bisect
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ofa-for-super-resolution_amd", "csrc")


def isa(src):
    out = os.path.join(tempfile.gettempdir(), "scan_" + os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
                        os.path.abspath(src)], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    return out


def scan(path):
    res, cur, body = [], None, []

    def flush():
        if cur is None:
            return
        waits = [i for i, l in enumerate(body) if "s_waitcnt vmcnt(0)" in l]
        loads = [i for i, l in enumerate(body) if re.search(r"\b(global|buffer|flat)_load", l)]
        n = 0
        for i in waits:
            j = bisect.bisect_right(loads, i)
            if j < len(loads) and loads[j] - i <= 30 and j > 0 and i - loads[j - 1] <= 30:
                n += 1
        res.append((n, len(loads), cur))

    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(line)
    flush()
    return res


def main():
    args = sys.argv[1:]
    least = int(args.pop(0)) if args and args[0].isdigit() else 2
    srcs = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    rows = []
    for src in srcs:
        rows += [(n, nl, k, os.path.basename(src)) for n, nl, k in scan(isa(src))]
    rows.sort(reverse=True)
    for n, nl, k, f in rows:
        if n >= least:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%3d serial / %3d loads  %s  [%s]" % (n, nl, name[:120], f))


if __name__ == "__main__":
    main()
