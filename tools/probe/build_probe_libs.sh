#!/bin/bash
# Instrumented builds of the library for the probes in this directory (not
# part of the product): same sources, one extra -D each, into tools/probe/.
#   libhk_mogpaths.so   -DHK_MOG_PATH_STATS   (mog_paths.py)
#   libhk_mogtimes.so   -DHK_MOG_WAVE_TIMES   (mog_timeline.py)
#   libhk_s3prof.so     -DHK_S3_PROF          (s3prof.py; asked for by name)
set -e
cd "$(dirname "$0")/../../hawkeye_amd/csrc"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -DHK_NO_DEVICE_SEEDS"
make -s
build() {   # name define [source]
  src=${3:-hk_mog.hip}
  mkdir -p _build/probe_$1
  $HIPCC $FLAGS -D$2 -x hip -c $src -o _build/probe_$1/$src.o
  objs=$(ls _build/*.o | grep -v "/$src.o")
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../tools/probe/libhk_$1.so _build/probe_$1/$src.o $objs
}
for which in "${@:-mogpaths mogtimes}"; do
  for w in $which; do
    case $w in
      mogpaths) build mogpaths HK_MOG_PATH_STATS ;;
      mogtimes) build mogtimes HK_MOG_WAVE_TIMES ;;
      s3prof) build s3prof HK_S3_PROF hk_arcs.cpp ;;
    esac
  done
done
