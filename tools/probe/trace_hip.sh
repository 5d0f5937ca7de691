#!/bin/bash
# HIP API trace (host side) + kernel trace of a pipelined batch: which API calls take milliseconds
set -o pipefail
OUT=$PWD/gpurun_out/trace_hip
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace --output-format csv -d "$OUT" -o h -- \
    python3 bench.py --steps 30 --warmup 2 --no-cpu-baseline --host-frames 0 --preroll-seconds 0.3 \
    > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 1
ls -la "$OUT"
python3 tools/probe/hip_calls.py "$OUT"/h_hip_api_trace.csv
