#!/bin/bash
# PMC passes over the thin-side conv kernels (one counter group per pass).  usage: bash tools/pmc_thin.sh OUTDIR "<kbench --only filter>" [dtype]
set -u
OUT=${1:-gpurun_out/pmc_ct}
ONLY=${2:-fwd 64->3}
DT=${3:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp -d "$R/$OUT/p$i" -o p --output-format csv -- python3 $R/tools/kbench.py --only "$ONLY" --dtype $DT --reps 6 > "$R/$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$R/$OUT/fail.log"
done
python3 - "$R/$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("KFILTER", "ct_") not in k:
            continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            fh.write("   %-34s n=%3d  mean %.5g  last %.5g\n" % (c, len(v), sum(v) / len(v), v[-1]))
print(open(out + "/summary.txt").read())
PY
