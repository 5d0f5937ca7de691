#!/bin/bash
# Thread / window scaling of the step-3 builder on the box's host cores.
export PYTHONPATH=$PWD
for t in ${THREADS:-14 16}; do for w in ${WINDOWS:-16 64 256}; do
  echo "T=$t W=$w"; HK_STEP3_WINDOW=$w HK_STEP3_STATS=1 HK_THREADS=$t python tools/time_step3.py tools/_events_C.npz 2>&1 | grep -E "step3:" | tail -2
done; done
