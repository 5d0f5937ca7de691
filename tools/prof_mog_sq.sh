#!/bin/bash
# SQ / GRBM counter passes of the shipped MOG batch kernel, alone on the GPU
# (tools/mog_prof.py), on the benchmark clip and on a static scene.  Run on the
# GPU box through gpurun from the repo root.  One rocprofv3 run per pass, at
# most 8 SQ_ counters each (TCC traffic counters in passes of their own), never
# combined with a trace mode.
set -o pipefail
TAG=${1:-r02}
CHUNK=${HK_PROF_CHUNK:-120}     # frames per launch: what the pipeline uses on a 1800-frame clip
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$PWD"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
P2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 GRBM_GUI_ACTIVE"
P4="FETCH_SIZE GRBM_GUI_ACTIVE"
P5="WRITE_SIZE GRBM_GUI_ACTIVE"
for scene in static clip; do
  i=0
  # the benchmark clip is profiled over all of its full launches (the mix of
  # update paths changes along the clip), the static scene over four
  if [ $scene = clip ]; then FRAMES=$((1800 / CHUNK * CHUNK)); else FRAMES=$((4 * CHUNK)); fi
  for pass in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    timeout -s KILL 150 rocprofv3 --pmc $pass --output-format csv \
        -d "$OUT/${scene}_p$i" -o pmc -- python3 tools/mog_prof.py --scene $scene --frames $FRAMES --chunk $CHUNK --reps 1 \
        > "$OUT/${scene}_p$i.log" 2> "$OUT/${scene}_p$i.err" || { echo "pass $scene $i failed"; tail -5 "$OUT/${scene}_p$i.err"; exit 1; }
  done
done
find "$OUT" -name '*.csv' | head
