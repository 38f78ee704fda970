#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03n
timeout 1200 python -m pytest tests/test_hem_gpu.py tests/test_configs_gpu.py tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r03n/t.log 2>&1; grep -E "passed|failed" gpurun_out/r03n/t.log
for r in 1 2; do python scripts/prof_hem.py 5000000 1 2 2>&1 | grep "rep1 L1 kernels"; done
python scripts/prof_hem.py 5000000 1 2 aniso 2>&1 | grep "rep1 L1 kernels"
