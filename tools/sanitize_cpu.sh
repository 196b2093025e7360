#!/bin/bash
# Round 6: AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side C / C++ of the repo (the GPU pool has no sanitizer runs):
#   oracle/liboracle.so (the checker), mvtn_amd/libmvr_hostmath.so (the product's per-pixel math built by g++), csrc/mvr_pack.cpp
#   (the host-side mesh packer, through a small driver) — run under the CPU tests that use them.  Builds the sanitized libraries IN
#   PLACE, runs, then rebuilds the ordinary ones (a trap does it on any exit).     bash tools/sanitize_cpu.sh
set -u
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
restore() {
  make -s -C oracle -B liboracle.so > /dev/null
  python3 -c "from mvtn_amd import build as b; b.build_hostmath(force=True)"
}
trap restore EXIT
gcc $SAN -fPIC -std=c99 -ffp-contract=off -fno-fast-math -fopenmp -shared -o oracle/liboracle.so oracle/mvr_oracle.c -lm || exit 1
g++ $SAN -std=c++17 -fPIC -shared -ffp-contract=off -o mvtn_amd/libmvr_hostmath.so mvtn_amd/csrc/mvr_hostmath.cpp -lm || exit 1
touch oracle/liboracle.so mvtn_amd/libmvr_hostmath.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
LD_PRELOAD="$ASAN $(gcc -print-file-name=libstdc++.so)" python3 -m pytest ${SAN_TESTS:-tests/test_oracle.py tests/test_host_logic.py} -q -s -m "not gpu" -p no:cacheprovider > /tmp/san_run.txt 2>&1; rc=$?; tail -40 /tmp/san_run.txt
echo "exit code of the sanitized test run: $rc"
# the host-side mesh packer as a library of its own (in the product it is linked into libmvr_hip.so); its error sink stubbed
printf 'extern "C" void mvr_set_error_(const char *) {}\n' > /tmp/mvr_pack_stub.cpp
g++ $SAN -std=c++17 -fPIC -shared -Iinclude -o /tmp/libmvr_pack_san.so mvtn_amd/csrc/mvr_pack.cpp /tmp/mvr_pack_stub.cpp || exit 1
LD_PRELOAD="$ASAN $(gcc -print-file-name=libstdc++.so)" python3 tools/sanitize_pack.py /tmp/libmvr_pack_san.so 2>&1 | tail -20
