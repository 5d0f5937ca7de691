#!/bin/bash
# Scratch: SQ / TA / TCC counter passes over tools/probe/mog_variant.py.
# usage: pmc_mog.sh <variant> <tag> [static]
set -o pipefail
V=$1; TAG=$2; ARG=$3
export TMPDIR=/tmp HK_MOG_VARIANT=$V
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
run() { name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o c -- python3 tools/probe/mog_variant.py $ARG > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  f=$(find $OUT/$name -name '*counter_collection.csv' | head -1)
  (head -1 $f; grep mog_batch2 $f) > $OUT/$name.csv
  rm -rf $OUT/$name
}
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM
run sq2 SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
run fetch FETCH_SIZE TCP_PENDING_STALL_CYCLES_sum
run write WRITE_SIZE
