#!/bin/bash
# Build libofasr_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${1:-libofasr_hip.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function"
objs=()
pids=()
for src in api.hip pixel_shuffle.hip ktransform.hip dwconv.hip pwconv.hip bnact.hip mbconv.hip conv2d.hip conv2d_f32.hip conv_thin.hip mbfused.hip resample.hip; do
    obj="${src%.hip}.o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ ofasr_common.h -nt "$obj" ] || [ ../../include/ofasr.h -nt "$obj" ]; then
        rm -f "$obj"                       # a failed compile must not leave a stale object to link
        hipcc $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
    objs+=("$obj")
done
for pid in "${pids[@]}"; do wait "$pid"; done   # `wait` without a PID ignores the jobs' exit codes
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $(pwd)/$OUT"
