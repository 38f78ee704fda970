#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_hem_gpu.py -x -q -m gpu 2>&1 | tail -15
for i in 1 2; do
echo "== iso fixed"; python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1 | cut -c1-60,180-420
echo "== iso exact"; GSR_HEM_PARTITION=exact python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1 | cut -c1-60,180-420
done
echo "== aniso"; python scripts/prof_hem.py 5000000 3 2 aniso 2>&1 | grep rep1 | cut -c1-60,180-420
