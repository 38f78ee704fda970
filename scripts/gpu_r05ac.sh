#!/bin/bash
# the next parent's pair list requested behind the present parent's last row loads: tests, A/B against the library without it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05ac; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hem_gpu.py -x -q > $OUT/tests.log 2>&1; echo "tests: exit $?"; grep -E "passed|failed|error" $OUT/tests.log | tail -3
for shape in iso aniso clustered; do bash scripts/ab_libs.sh $shape - noah 2>&1 | tee -a $OUT/ab_mstep_pairs_ahead.txt; done
