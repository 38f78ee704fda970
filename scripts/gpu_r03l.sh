#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03l
for v in walk staged; do
  export GSR_HEM_PARTITION=$v
  for r in 1 2; do python scripts/prof_hem.py 5000000 1 2 2>&1 | grep "rep1 L1 kernels"; done
done
