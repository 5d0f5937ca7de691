#!/bin/bash
# Pipelined throughput of bench.py over (clips in step 3 at a time, worker
# threads per clip): how should a rank split its host cores?
out=gpurun_out/sweep_builders.log
: > $out
for cfg in ${CFGS:-"2 7" "4 3" "5 3" "7 2" "10 1" "14 1" "16 1"}; do
  set -- $cfg
  echo "builders=$1 threads=$2 steps=${STEPS:-20}" >> $out
  timeout -k 10 150 python bench.py --steps ${STEPS:-20} --builders $1 --host-threads $2 \
      --no-cpu-baseline --host-frames 0 2>>gpurun_out/sweep_builders.err | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['result']['arcs'])" >> $out || exit 1
done
cat $out
