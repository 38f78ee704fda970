#!/bin/bash
# CPU-side sanitizer runs of the oracle (build container; no GPU involved):
#   1. the driver program under ASan + UBSan and under TSan (OpenMP through libomp + Archer)
#   2. the golden-vector and oracle unit tests with the ASan/UBSan build of the library preloaded into python
set -euo pipefail
cd "$(dirname "$0")/.."
make -C oracle asan tsan > /dev/null
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
./oracle/sanitize_asan 4 6000
./oracle/sanitize_asan 1 2000
LLVM=${LLVM:-/opt/rocm/lib/llvm}
# Archer teaches TSan OpenMP's synchronisation; without it every barrier of the (uninstrumented) runtime is a false positive
export TSAN_OPTIONS="ignore_noninstrumented_modules=1 halt_on_error=1" OMP_TOOL_LIBRARIES=$LLVM/lib/libarcher.so ARCHER_OPTIONS="verbose=0"
./oracle/sanitize_tsan 4 4000
unset TSAN_OPTIONS OMP_TOOL_LIBRARIES
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" GSR_ORACLE_LIB=$PWD/oracle/libgsr_oracle_asan.so \
    python -m pytest -q -x tests/test_oracle_golden.py tests/test_oracle_math.py tests/test_rng.py -p no:cacheprovider
echo "sanitizers: clean"
