#!/bin/bash
# kernel trace of a pipelined batch alone (no single-clip pass afterwards would be nice; the
# analysis cuts the batch out by time): gpurun_out/trace_batch/
set -o pipefail
OUT=$PWD/gpurun_out/trace_batch
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o b -- \
    python3 bench.py --steps ${STEPS:-20} --warmup 2 --no-cpu-baseline --host-frames 0 \
    > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 1
python3 tools/probe/trace_gaps.py "$OUT"/b_kernel_trace.csv 15
