#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -k "asynchronous or zero_copy or sh_rows_read" > $OUT/tests_erase.log 2>&1; echo "erase tests: exit $?"; tail -4 $OUT/tests_erase.log
for SHAPE in aniso iso; do echo "== $SHAPE"; GSR_HEM_TIMING=1 python scripts/prof_hem.py 5000000 1 5 $SHAPE 2>&1 | grep "L1" | tail -4 | cut -c1-330; done 2>&1 | tee $OUT/level_5m.txt
timeout 2400 python -m pytest tests/test_hem_gpu.py tests/test_configs_gpu.py tests/test_stress_gpu.py -q > $OUT/hem_tests.log 2>&1; echo "hem+configs+stress: exit $?"; tail -3 $OUT/hem_tests.log
