#!/bin/bash
# Round profile set (run on the GPU box through gpurun): for the bf16 AND the fp32 training step of `bench.py` the
# rocprofv3 kernel trace + stats of the timed steps, per-stream balance, and the two PMC passes for HBM traffic
# (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, as MI355X_MICROARCH.md prescribes); then the kernel trace (+ PMC passes) of
# the config-5 evaluation.  Results land in gpurun_out/<tag>_*; copy what is to be judged into profiles/
# (trace_summary.json and pmc_traffic.json are what bench.py's roofline.trace_avg_us / .traffic read).
#   usage: bash tools/prof_round.sh <tag> [parts]      parts: any of b (bf16) f (fp32) e (config 5), default bfe
set -uo pipefail
TAG=${1:-r03}
PARTS=${2:-bfe}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-roofline --no-fp32 --no-cpu-baseline"
SUMM=$OUT/${TAG}_trace_summary.json
PMCJ=$OUT/${TAG}_pmc_traffic.json
rm -f "$SUMM" "$PMCJ"

leg() {   # leg <dtype> <suffix> <steps> <warmup> <steps aggregated>
    local DT=$1 SFX=$2 ST=$3 WU=$4 AGG=$5
    rm -rf /tmp/pk /tmp/pf /tmp/pw
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o p -- python3 $R/bench.py --dtype $DT --steps $ST --warmup $WU $COMMON > $OUT/${TAG}${SFX}_prof_bench.json 2> $OUT/${TAG}${SFX}_prof.err
    local TR=$(find /tmp/pk -name "*kernel_trace.csv" | head -1)
    python3 $R/profiles/trace_steps.py "$TR" $AGG 80 0 "$SUMM" $DT > $OUT/${TAG}${SFX}_steps.txt
    python3 $R/tools/stream_balance.py "$TR" $AGG 0 > $OUT/${TAG}${SFX}_streams.txt
    cp "$(find /tmp/pk -name "*kernel_stats.csv" | head -1)" $OUT/${TAG}${SFX}_kernel_stats.csv
    echo "$DT trace done" >> $OUT/${TAG}_prof.log
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --dtype $DT --steps 5 --warmup 3 $COMMON > /dev/null 2>> $OUT/${TAG}${SFX}_prof.err
    # the counter pass serializes the dispatches: its kernel trace = every kernel's duration WITHOUT the other stream
    # beside it (trace_summary.json key <dtype>_serialized; bench.py: roofline.serialized_avg_us / frac_serialized)
    python3 $R/profiles/trace_steps.py "$(find /tmp/pf -name "*kernel_trace.csv" | head -1)" 3 80 0 "$SUMM" ${DT}_serialized > $OUT/${TAG}${SFX}_serialized_steps.txt
    echo "$DT fetch done" >> $OUT/${TAG}_prof.log
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --dtype $DT --steps 5 --warmup 3 $COMMON > /dev/null 2>> $OUT/${TAG}${SFX}_prof.err
    python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw "$PMCJ" --merge > $OUT/${TAG}${SFX}_pmc_traffic.txt
    echo "$DT pmc done" >> $OUT/${TAG}_prof.log
    head -12 $OUT/${TAG}${SFX}_steps.txt | cut -c1-150; cat $OUT/${TAG}${SFX}_streams.txt
}

case $PARTS in *b*) leg bf16 "" 12 6 8 ;; esac
case $PARTS in *f*) leg f32 _f32 8 4 6 ;; esac
case $PARTS in *e*)
    rm -rf /tmp/pe /tmp/pf /tmp/pw
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -o e -- python3 $R/bench.py --config c5 --dtype bf16 --steps 6 --warmup 2 --no-roofline > $OUT/${TAG}_prof_bench_c5.json 2>> $OUT/${TAG}_prof.err
    cp "$(find /tmp/pe -name "*kernel_stats.csv" | head -1)" $OUT/${TAG}_c5_kernel_stats.csv
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/bench.py --config c5 --steps 4 --warmup 2 --no-roofline > /dev/null 2>> $OUT/${TAG}_prof.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/bench.py --config c5 --steps 4 --warmup 2 --no-roofline > /dev/null 2>> $OUT/${TAG}_prof.err
    python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw "$PMCJ" --merge > $OUT/${TAG}_c5_pmc_traffic.txt
    echo "c5 done" >> $OUT/${TAG}_prof.log ;;
esac
