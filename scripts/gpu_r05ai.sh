#!/bin/bash
# Onesweep with 12 keys per thread (default now) against 16 (rounds 1-5) and 10, over the three level sizes and the three shapes
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05ai; mkdir -p $OUT
for shape in iso aniso; do bash scripts/ab_libs.sh $shape - ipt16 ipt10 2>&1 | tee -a $OUT/ab_sort_ipt.txt; done
