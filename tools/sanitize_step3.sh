#!/bin/bash
# Sanitizers over the host side of step 3 (speculative multi-thread builder, helpers that join
# from outside): hk_common.cpp, hk_arcs.cpp and hk_simd.cc are built with the sanitizer on the
# host side only (GPU sanitizers are not available on this pool) and libstdc++'s assertions
# (bounds-checked operator[]), linked with the library's other objects and a small driver
# (tools/sanitize_step3_driver.cpp) that builds the arcs of a synthetic tag set with N own
# threads while M outside threads call hk_step3_help.  No GPU needed.
#   tools/sanitize_step3.sh thread     ThreadSanitizer
#   tools/sanitize_step3.sh address    AddressSanitizer + UndefinedBehaviorSanitizer
# Prints the builds and every sanitizer report.
set -e
cd "$(dirname "$0")/.."
KIND=${1:-thread}
case $KIND in
  thread)  HOST="-Xarch_host -fsanitize=thread";  PLAIN="-fsanitize=thread" ;;
  address) HOST="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined"
           PLAIN="-fsanitize=address -fsanitize=undefined" ;;
  *) echo "usage: $0 thread|address"; exit 2 ;;
esac
make -s -C hawkeye_amd/csrc
OUT=${TMPDIR:-/tmp}/hk_sanitize_$KIND
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FL="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DHK_NO_DEVICE_SEEDS -D_GLIBCXX_ASSERTIONS -Iinclude"
for f in hk_common.cpp hk_arcs.cpp; do
  $HIPCC $FL $HOST -x hip -c hawkeye_amd/csrc/$f -o "$OUT/$f.o"
done
$HIPCC -x c++ -O1 -g -std=c++17 -fPIC -mavx2 -ffp-contract=off $PLAIN \
    -c hawkeye_amd/csrc/hk_simd.cc -o "$OUT/hk_simd.o"
OBJS=$(ls hawkeye_amd/csrc/_build/*.hip.o)
$HIPCC --offload-arch=gfx950 $HOST -Iinclude -x hip tools/sanitize_step3_driver.cpp \
    -x none "$OUT"/hk_common.cpp.o "$OUT"/hk_arcs.cpp.o "$OUT"/hk_simd.o $OBJS -o "$OUT/drv" $PLAIN
echo "# $KIND sanitizer, host side of step 3"
for cfg in "8 0" "1 5" "2 4" "4 3"; do
  echo "== own threads / helper threads: $cfg"
  TSAN_OPTIONS="halt_on_error=0" ASAN_OPTIONS="detect_leaks=0" "$OUT/drv" $cfg 2>&1 |
      grep -E "^threads|Sanitizer|runtime error|Assertion|SUMMARY" | sort | uniq -c
done
