#!/bin/bash
# bucketed pair copy as {16-bit slot, wL}: HEM + distributed tests, then the level figures
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05q; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_hem_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_modes_gpu.py -x -q > $OUT/tests.log 2>&1; echo "tests: exit $?"; grep -E "passed|failed|error" $OUT/tests.log | tail -3
for i in 1 2 3; do python scripts/prof_hem.py 5000000 3 3 iso 2>&1 | grep 'rep2 L. kernels' | cut -c1-200; done | tee $OUT/levels.txt
