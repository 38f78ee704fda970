#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for round in 1 2 3; do for v in ob10 ob7 ob6; do
  echo "$v: $(GSR_HIP_LIB=$PWD/variants/$v.so timeout 200 python scripts/prof_hem.py 5000000 1 2 2>&1 | grep 'rep1 L1 kernels')"
done; done
for v in ob10 ob7 ob6; do
  echo "$v 1.67M: $(GSR_HIP_LIB=$PWD/variants/$v.so timeout 200 python scripts/prof_hem.py 1670000 1 3 2>&1 | grep 'rep2 L1 kernels')"
done
