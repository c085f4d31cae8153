#!/bin/bash
# Round profile set (run on the GPU box through gpurun): rocprofv3 kernel trace + stats of the timed training steps,
# per-stream balance, the two PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as
# MI355X_MICROARCH.md prescribes), and the kernel trace of the config-5 evaluation.  Results land in gpurun_out/<tag>_*;
# copy what is to be judged into profiles/.   usage: bash tools/prof_round.sh <tag>
set -uo pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 12 --warmup 6 --no-roofline --no-fp32 --no-cpu-baseline"
rm -rf /tmp/pk /tmp/pf /tmp/pw /tmp/pe
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python3 $R/bench.py $ARGS > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err
TR=$(find /tmp/pk -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/pk -name "*kernel_stats.csv" | head -1)
python3 $R/profiles/trace_steps.py "$TR" 8 80 0 > $OUT/${TAG}_steps.txt
python3 $R/tools/stream_balance.py "$TR" 8 0 > $OUT/${TAG}_streams.txt
cp "$ST" $OUT/${TAG}_kernel_stats.csv
echo "trace done" >> $OUT/${TAG}_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --steps 6 --warmup 3 --no-roofline --no-fp32 --no-cpu-baseline > /dev/null 2>> $OUT/${TAG}_prof.err
echo "fetch done" >> $OUT/${TAG}_prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --steps 6 --warmup 3 --no-roofline --no-fp32 --no-cpu-baseline > /dev/null 2>> $OUT/${TAG}_prof.err
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
echo "pmc done" >> $OUT/${TAG}_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o e -- python3 $R/bench.py --config c5 --dtype bf16 --steps 6 --warmup 2 --no-roofline > $OUT/${TAG}_prof_bench_c5.json 2>> $OUT/${TAG}_prof.err
cp "$(find /tmp/pe -name "*kernel_stats.csv" | head -1)" $OUT/${TAG}_c5_kernel_stats.csv
head -3 $OUT/${TAG}_steps.txt; cat $OUT/${TAG}_streams.txt
