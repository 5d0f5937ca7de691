#!/bin/bash
# Thread / batch scaling of the step-3 builder on the box's host cores.
export PYTHONPATH=$PWD
for t in 1 4 8 16; do for b in 16 32 64; do
  echo "T=$t B=$b"; HK_STEP3_WINDOW=$b HK_STEP3_STATS=1 HK_THREADS=$t python tools/time_step3.py tools/_events_C.npz 2>&1 | grep -E "step3:" | tail -2
done; done
