#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root.  Kernel trace +
# stats of the default bench under gpurun_out/prof_<tag>/ .  The counter
# passes (SQ, FETCH_SIZE, WRITE_SIZE) are tools/prof_mog_sq.sh: counters are
# never combined with a trace mode (gpurun refuses that).
# Afterwards, here:  cp gpurun_out/prof_<tag>/trace/bench_kernel_stats.csv
#                       profiles/<tag>_kernel_stats.csv   (and the bench line)
set -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT/trace" -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || exit 1
find "$OUT" -name '*.csv' | head -20
