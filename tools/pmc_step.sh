#!/bin/bash
# PMC passes (one counter group per pass) over kernels of the bench step.  usage: KFILTER=<kernel substring> bash tools/pmc_step.sh OUTDIR [bench.py args]
# (under --pmc the dispatches are serialized: the numbers are each kernel's own, without the other stream beside it)
set -u
OUT=${1:-gpurun_out/pmc_step}
shift || true
ARGS=${*:---no-fp32 --steps 3 --warmup 2 --no-roofline --no-cpu-baseline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp -d "$R/$OUT/p$i" -o p --output-format csv -- python3 $R/bench.py $ARGS > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$R/$OUT/fail.log"
done
python3 - "$R/$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("KFILTER", "pw_fanin_pipe") not in k:
            continue
        agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("KFILTER", "pw_fanin_pipe") in k:
            dur[k[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(agg):
        d = dur.get(k, [])
        fh.write("%s   launches %d  avg %.1f us (serialized)\n" % (k, len(d), sum(d) / max(len(d), 1)))
        for c in sorted(agg[k]):
            v = agg[k][c]
            fh.write("   %-34s n=%3d  mean %.5g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
rm -rf "$R/$OUT"/p[0-9]*   # the raw counter / trace CSVs of a whole bench run are tens of MB per pass; summary.txt stays
