#!/bin/bash
# M-step fetching ahead for the wave's next parent: HEM tests, then A/B against the library without it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05p; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hem_gpu.py -x -q > $OUT/hem_tests.log 2>&1; echo "hem tests: exit $?"; grep -E "passed|failed|error" $OUT/hem_tests.log | tail -3
for shape in iso aniso clustered; do bash scripts/ab_libs.sh $shape - noahead aheadU2 2>&1 | tee -a $OUT/ab_mstep_ahead.txt; done
