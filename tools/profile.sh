#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root.  Writes under
# gpurun_out/prof_<tag>/ : a kernel-trace + stats run of the default bench
# and two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a short run.
# Counters are never combined with a trace mode (gpurun refuses that).
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT/trace" -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err" || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv \
    -d "$OUT/pmc_fetch" -o pmc -- python3 bench.py --steps 1 --warmup 0 --frames 384 --no-cpu-baseline \
    > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv \
    -d "$OUT/pmc_write" -o pmc -- python3 bench.py --steps 1 --warmup 0 --frames 384 --no-cpu-baseline \
    > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 3
find "$OUT" -name '*.csv' | head -20
