#!/bin/bash
# k_join_parts ahead of the SH copy's fork, four orphan rows in flight: tests + level figures on the three shapes
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05s; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_hem_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_modes_gpu.py tests/test_configs_gpu.py -x -q > $OUT/tests.log 2>&1; echo "tests: exit $?"; grep -E "passed|failed|error" $OUT/tests.log | tail -3
for shape in iso aniso clustered; do for i in 1 2 3; do python scripts/prof_hem.py 5000000 1 3 $shape 2>&1 | grep 'rep2 L. kernels' | cut -c1-200; done; done | tee $OUT/levels.txt
GSR_HEM_TIMING=1 bash scripts/gpu_kstats.sh 5000000 aniso 2>&1 | grep -E "k_orphan_rows|k_join_parts|k_erase|rep2 L1 kernels"
GSR_HEM_TIMING=1 bash scripts/gpu_kstats.sh 5000000 iso 2>&1 | grep -E "k_orphan_rows|k_join_parts|k_mstep_headers|rep2 L1 kernels"
