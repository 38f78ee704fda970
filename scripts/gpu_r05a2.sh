#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 2400 python scripts/fullsize_modes.py c5 --splats 40000000 --world 8 --out $OUT/r05_c5_40m_8ranks.json > $OUT/c5.log 2>&1; echo "c5 full size: exit $?"; tail -1 $OUT/c5.log | cut -c1-600
timeout 900 python -m pytest tests/test_fullsize_modes_gpu.py -x -q > $OUT/test_fullsize.log 2>&1; echo "fullsize tests: exit $?"; tail -3 $OUT/test_fullsize.log
