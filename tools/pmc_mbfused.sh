#!/bin/bash
# PMC passes over the fused eval-mode block kernel (one counter group per pass; --kernel-trace only).
# usage (on the GPU box): bash tools/pmc_mbfused.sh OUTDIR
set -u
OUT=${1:-gpurun_out/pmc_mf}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $grp -d "$OUT/p$i" -o p --output-format csv -- python3 tools/probe_mbfused.py 64x64 > "$OUT/p$i.log" 2>&1 || echo "pass $i failed" >> "$OUT/fail.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mb_fused_kernel" not in k:
            continue
        agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k in sorted(agg):
        fh.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            fh.write("   %-34s n=%3d  mean %.4g  last %.4g\n" % (c, len(v), sum(v) / len(v), v[-1]))
print(open(out + "/summary.txt").read())
PY
