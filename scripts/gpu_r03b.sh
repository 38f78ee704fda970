#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r03b; mkdir -p $OUT
python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "stage1 or anisotropic or irregular or clipping or golden or known_answer or fast_log or sweep or outliers" 2>&1 | tail -15
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "anisotropic" 2>&1 | tail -5
echo "== iso"; python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1
echo "== aniso"; python scripts/prof_hem.py 5000000 3 2 aniso 2>&1 | grep rep1
