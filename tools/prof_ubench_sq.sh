#!/bin/bash
# Calibration of the SQ counters used for the VALU roofline: the issue
# microbenchmark (pure streams of one instruction kind at fixed residency)
# under the same counters as tools/prof_mog_sq.sh pass 1.
set -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/sq_ubench_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p1" -o pmc -- tools/ubench/valu_issue > "$OUT/p1.log" 2> "$OUT/p1.err" || { echo "ubench pass failed"; tail -5 "$OUT/p1.err"; exit 1; }
find "$OUT" -name '*.csv' | head
