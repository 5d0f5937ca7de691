#!/bin/bash
# ThreadSanitizer over the host side of step 3 (speculative multi-thread builder, helpers that
# join from outside): hk_common.cpp and hk_arcs.cpp are built with -fsanitize=thread on the host
# side only (GPU sanitizers are not available on this pool), linked with the library's other
# objects and a small driver (tools/tsan_step3_driver.cpp) that builds the arcs of a synthetic
# tag set with N own threads while M outside threads call hk_step3_help.  No GPU needed.
#   tools/tsan_step3.sh            -> prints the builds and every ThreadSanitizer report
set -e
cd "$(dirname "$0")/.."
make -s -C hawkeye_amd/csrc
OUT=${TMPDIR:-/tmp}/hk_tsan
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FL="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DHK_NO_DEVICE_SEEDS -Iinclude"
for f in hk_common.cpp hk_arcs.cpp; do
  $HIPCC $FL -Xarch_host -fsanitize=thread -x hip -c hawkeye_amd/csrc/$f -o "$OUT/$f.o"
done
$HIPCC -x c++ -O1 -g -std=c++17 -fPIC -mavx2 -ffp-contract=off -fsanitize=thread \
    -c hawkeye_amd/csrc/hk_simd.cc -o "$OUT/hk_simd.o"
OBJS=$(ls hawkeye_amd/csrc/_build/*.hip.o)
$HIPCC --offload-arch=gfx950 -Xarch_host -fsanitize=thread -Iinclude -x hip tools/tsan_step3_driver.cpp \
    -x none "$OUT"/hk_common.cpp.o "$OUT"/hk_arcs.cpp.o "$OUT"/hk_simd.o $OBJS -o "$OUT/drv" -fsanitize=thread
for cfg in "8 0" "1 5" "2 4" "4 3"; do
  echo "== own threads / helper threads: $cfg"
  TSAN_OPTIONS="halt_on_error=0" "$OUT/drv" $cfg 2>&1 | grep -E "^threads|ThreadSanitizer|SUMMARY" | sort | uniq -c
done
