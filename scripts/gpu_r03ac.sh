#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for round in 1 2; do for v in bs1024 bs512; do
  echo "$v: $(GSR_HIP_LIB=$PWD/variants/$v.so timeout 200 python scripts/prof_hem.py 5000000 1 2 2>&1 | grep 'rep1 L1 kernels')"
done; done
