#!/bin/bash
# tools/sanitize_cpu.sh -- AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU side: the oracle
# (oracle/*.cpp), the CPU baseline (oracle/cpu_fast.cpp) and the host build of the kernel arithmetic (tests/hostsim), through
# the CPU test-suite's own cases.  GPU sanitizers are not available on this pool; the kernels' arithmetic is the same
# ssde_math.hpp / ssde_tv.hpp that tests/hostsim compiles for the host.     bash tools/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/build/san"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -pthread"
g++ $SAN -shared -o "$OUT/liboracle.so" "$ROOT/oracle/oracle_capi.cpp"
g++ $SAN -shared -o "$OUT/liboracle_quad.so" "$ROOT/oracle/oracle_quad.cpp" -lquadmath
g++ $SAN -mfma -mavx2 -shared -o "$OUT/libcpu_fast.so" "$ROOT/oracle/cpu_fast.cpp"
g++ $SAN -shared -o "$OUT/libhostsim.so" "$ROOT/tests/hostsim/hostsim.cpp"
ASAN_LIB="$(g++ -print-file-name=libasan.so)"
cd "$ROOT"
# (python itself is not instrumented: leak reports about the interpreter are noise)
LD_PRELOAD="$ASAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  SSDE_ORACLE_LIBDIR="$OUT" python -m pytest tests/test_oracle_golden.py tests/test_oracle_quad.py tests/test_kernel_math_host.py tests/test_laplace.py \
  -q -m "not gpu" -p no:cacheprovider "$@"
