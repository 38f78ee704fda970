#!/bin/bash
# HEM / partition / multi-process tests and one-level timings at 200 k, 556 k and 5 M splats (through gpurun)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/hem_check
timeout 1500 python -m pytest tests/test_hem_gpu.py tests/test_configs_gpu.py tests/test_distributed_gpu.py tests/test_abi.py -x -q -m gpu > gpurun_out/hem_check/t.log 2>&1; grep -E "passed|failed" gpurun_out/hem_check/t.log
for n in 200000 556000 5000000; do python scripts/prof_hem.py $n 1 3 2>&1 | grep "rep2 L1 kernels" | sed -e "s/^/n=$n /"; done
