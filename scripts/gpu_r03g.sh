#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for round in 1 2; do
  for v in k1 k4 k4w6 k4w7 k4s0 k4s0w7 k2 k2w7; do
    echo "round $round $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python scripts/prof_hem.py 5000000 2 2 iso 2>&1 | grep 'rep1' | sed -e 's/.*pairs/pairs/' | cut -c1-30,150-330 | tr '\n' '|')"
  done
done
for v in k1 k4 k4s0; do echo "aniso $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python scripts/prof_hem.py 5000000 1 2 aniso 2>&1 | grep 'rep1' | sed -e 's/.*pairs/pairs/' | cut -c1-30,150-330)"; done
# correctness of the grouped kernel: a few HEM tests with the k4 variant
GSR_HIP_LIB=$PWD/variants/k4.so python -m pytest tests/test_hem_gpu.py -x -q -m gpu -k "golden or known_answer or cascade or heavy or two_pass or anisotropic or irregular or sweep or partition" 2>&1 | tail -4
