#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE on the access shapes of mog_batch2_k
# (tools/ubench/hbm_shapes.hip): one rocprofv3 --pmc pass per case and
# counter, the program itself directly after `--`.  Run on the GPU box from
# the repo root; results under gpurun_out/hbm_shapes_<tag>/.
set -o pipefail
TAG=${1:-r03}
OUT=$PWD/gpurun_out/hbm_shapes_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for c in lines16 seg8 seg12 plane8r; do
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${c}_fetch" -o pmc \
      -- tools/ubench/hbm_shapes $c > "$OUT/${c}_fetch.log" 2> "$OUT/${c}_fetch.err" || { echo "$c fetch failed"; tail -3 "$OUT/${c}_fetch.err"; exit 1; }
done
for c in lines16w plane8w seg8; do
  timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${c}_write" -o pmc \
      -- tools/ubench/hbm_shapes $c > "$OUT/${c}_write.log" 2> "$OUT/${c}_write.err" || { echo "$c write failed"; tail -3 "$OUT/${c}_write.err"; exit 1; }
done
for c in lines16 seg8 plane8r; do
  timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/${c}_tcc" -o pmc \
      -- tools/ubench/hbm_shapes $c > "$OUT/${c}_tcc.log" 2> "$OUT/${c}_tcc.err" || { echo "$c tcc failed"; tail -3 "$OUT/${c}_tcc.err"; }
done
find "$OUT" -name '*counter_collection.csv' | sort
