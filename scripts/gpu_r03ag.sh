#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "partition or oracle or large" 2>&1 | grep -E "passed|failed"
for n in 5000000 8000000 20000000 40000000; do timeout 600 python scripts/prof_hem.py $n 1 2 2>&1 | grep -E "rep1 L1 kernels|rror" | sed -e "s/^/n=$n /"; done
