#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for v in abl1 abl2 abl3; do
  for r in 1 2; do GSR_HIP_LIB=$PWD/variants/$v.so python scripts/prof_hem.py 5000000 1 2 2>&1 | grep "rep1 L1 kernels" | sed -e "s/^/$v /"; done
done
