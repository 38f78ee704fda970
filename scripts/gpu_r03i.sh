#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "partitioned_level_with_one_rank" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "spatially" 2>&1 | tail -25
