#!/bin/bash
# Throughput of M independent single-threaded step-3 builds side by side
# (is the 16-thread builder limited by the cores or by its own structure?).
export PYTHONPATH=$PWD
python -c "import hawkeye_amd.arcs" || exit 1
for m in ${PROCS:-1 4 8 14 16}; do
  echo "M=$m T=${T:-1}"
  pids=()
  for i in $(seq $m); do
    HK_THREADS=${T:-1} python tools/time_step3.py tools/_events_C.npz > /tmp/s3p_$i.log 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  for i in $(seq $m); do tail -1 /tmp/s3p_$i.log; done | sort | uniq -c
done
