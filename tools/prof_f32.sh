#!/bin/bash
# rocprofv3 kernel trace of the fp32 training step (the reference's arithmetic).  usage: bash tools/prof_f32.sh <tag>
set -uo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk32 -o p -- python3 $R/bench.py --dtype f32 --steps 8 --warmup 4 --no-roofline --no-cpu-baseline > $OUT/${TAG}_f32_prof_bench.json 2> $OUT/${TAG}_f32_prof.err
TR=$(find /tmp/pk32 -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/pk32 -name "*kernel_stats.csv" | head -1)
python3 $R/profiles/trace_steps.py "$TR" 6 80 0 > $OUT/${TAG}_f32_steps.txt
python3 $R/tools/stream_balance.py "$TR" 6 0 > $OUT/${TAG}_f32_streams.txt
cp "$ST" $OUT/${TAG}_f32_kernel_stats.csv
head -40 $OUT/${TAG}_f32_steps.txt; cat $OUT/${TAG}_f32_streams.txt
