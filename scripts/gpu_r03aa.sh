#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for round in 1 2; do
for v in rb8 rb10 rb11; do
  echo "$v: $(GSR_HIP_LIB=$PWD/variants/$v.so timeout 200 python scripts/prof_hem.py 5000000 1 2 2>&1 | grep 'rep1 L1: wall' | sed -e 's/.*ms_grid/ms_grid/' | cut -c1-120)"
done
done
GSR_HIP_LIB=$PWD/variants/rb10.so timeout 600 python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "equals_oracle or known_answers or anisotropic" 2>&1 | grep -E "passed|failed"
