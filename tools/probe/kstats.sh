#!/bin/bash
# Scratch: kernel-trace stats of a probe script. usage: kstats.sh <tag> <script> [args]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python3 "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), 'us avg', r['Percentage'])
P
find $OUT -name '*kernel_trace.csv' -delete
