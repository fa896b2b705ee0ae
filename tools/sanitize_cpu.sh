#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU code of the repo (GPU sanitizers are not available on the pool):
#   the oracle (oracle/se3d_oracle.cpp, loaded by the CPU tests through ctypes) and the CPU-only host drivers (time synchroniser, markers).
# Builds the instrumented binaries in place, runs the CPU tests that use them, restores the normal builds.  Run from the repo root.
set -eu
cd "$(dirname "$0")/.."
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
ASAN_LIB="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
trap 'make -s -B -C oracle libse3d_oracle.so; make -s -B -C host se3d_sync_test se3d_markers_test' EXIT
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -fopenmp -shared -o oracle/libse3d_oracle.so oracle/se3d_oracle.cpp
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0 python -m pytest -q -s \
  tests/test_oracle_kat.py tests/test_oracle_branches.py tests/test_oracle_dlt_cpu.py tests/test_oracle_stage2.py tests/test_oracle_hungarian.py \
  tests/test_oracle_reproject.py tests/test_oracle_regression.py tests/test_harness_cpu.py 2>&1 | tee /tmp/sanitize_oracle.log | tail -3
if grep -q "runtime error\|AddressSanitizer" /tmp/sanitize_oracle.log; then echo "SANITIZER FINDINGS (oracle): /tmp/sanitize_oracle.log"; exit 1; fi
make -s -B -C host CXXFLAGS="$SAN -std=c++17 -fPIC -Wall -Wextra -fno-sanitize-recover=undefined" se3d_sync_test se3d_markers_test
python -m pytest -q tests/test_sync_cpu.py tests/test_markers_cpu.py | tail -2
# round 6: the staged all-gather's rendezvous of the C++ exchange (threads as ranks, plain host memory as device memory)
g++ $SAN -std=c++17 -fPIC -Wall -Wextra -fno-sanitize-recover=undefined -o /tmp/se3d_exchange_test_san host/se3d_exchange_test.cpp host/se3d_exchange.cpp \
  -Lsmartedgesensor3dhumanpose_amd -lse3d_hip -ldl -lpthread -Wl,-rpath,"$PWD/smartedgesensor3dhumanpose_amd"
ASAN_OPTIONS=detect_leaks=0 /tmp/se3d_exchange_test_san 5 9 2>&1 | tee /tmp/sanitize_exchange.log | tail -2
if grep -q "runtime error\|AddressSanitizer" /tmp/sanitize_exchange.log; then echo "SANITIZER FINDINGS (exchange): /tmp/sanitize_exchange.log"; exit 1; fi
echo "sanitize_cpu: clean"
