#!/bin/bash
# Kernel trace and SQ counter passes of the blob kernels alone on the GPU
# (tools/blob_prof.py).  Run on the GPU box through gpurun from the repo root.
set -o pipefail
TAG=${1:-r03}
CHUNK=${2:-360}     # frames per detection batch (ClipScanner's default)
NO=${3:-2}          # which chunk of the clip (0-based): past the learning ramp
OUT=$PWD/gpurun_out/blob_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o blob \
    -- python3 tools/blob_prof.py --chunk $CHUNK --chunk-no $NO > "$OUT/trace.log" 2> "$OUT/trace.err" || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p1" -o pmc -- python3 tools/blob_prof.py --chunk $CHUNK --chunk-no $NO --reps 3 > "$OUT/p1.log" 2> "$OUT/p1.err" || exit 2
cat "$OUT/trace.log"
