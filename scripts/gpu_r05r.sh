#!/bin/bash
# k_mstep_headers without its per-wave same-address atomic, k_join_parts with eight pairs in flight, halo-stage errors agreed on: tests + levels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05r; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_hem_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_modes_gpu.py -x -q > $OUT/tests.log 2>&1; echo "tests: exit $?"; grep -E "passed|failed|error" $OUT/tests.log | tail -3
for shape in iso clustered; do for i in 1 2 3; do python scripts/prof_hem.py 5000000 3 3 $shape 2>&1 | grep 'rep2 L. kernels' | cut -c1-200; done; done | tee $OUT/levels.txt
python scripts/level_ladder.py 2>&1 | tail -12 | tee $OUT/ladder.txt
