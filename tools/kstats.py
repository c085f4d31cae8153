#!/usr/bin/env python3
"""Print (kernel, calls, average us) from a rocprofv3 *_kernel_stats.csv, optionally filtered by substring."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if pat in r["Name"]:
        print("%-70s calls=%6s avg_us=%9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
