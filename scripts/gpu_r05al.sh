#!/bin/bash
# the large-input scan configuration from 2^20 elements on (default), from 2^18 on, never: three level sizes
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05al; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q > $OUT/tests.log 2>&1; echo "tests: exit $?"; grep -E "passed|failed" $OUT/tests.log | tail -2
bash scripts/ab_libs.sh iso - scanbig256k scanbignever 2>&1 | tee $OUT/ab_scan_big.txt
