#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
timeout 600 python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "partition or equals_oracle or anisotropic" 2>&1 | tail -5
bash scripts/ab_env.sh 5000000 1 GSR_HEM_PARTITION walk staged
export GSR_HEM_SH_OVERLAP=2
bash scripts/ab_env.sh 5000000 1 GSR_HEM_SH_GRID 0 256 512 1024
