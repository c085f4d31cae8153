#!/bin/bash
# Per-kernel durations of the bench step with the dispatches SERIALIZED (rocprofv3 --pmc serializes them): each kernel's own
# time without the other stream beside it, to set against the concurrent trace of tools/prof_round.sh.
# usage: bash tools/isolated_steps.sh OUT.txt [bench.py args]
set -u
OUTF=${1:-gpurun_out/isolated_steps.txt}
shift || true
ARGS=${*:---dtype bf16 --no-fp32 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/iso_prof
rocprofv3 --kernel-trace --pmc SQ_WAVES -d /tmp/iso_prof -o p --output-format csv -- python3 $R/bench.py $ARGS > /tmp/iso_prof.log 2>&1
TR=$(find /tmp/iso_prof -name "*kernel_trace.csv" | head -1)
python3 $R/profiles/trace_steps.py "$TR" 3 80 0 > "$R/$OUTF"
head -3 "$R/$OUTF"
