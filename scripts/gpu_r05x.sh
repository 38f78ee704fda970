#!/bin/bash
# the SH copy forked in front of k_select, by a few workgroups on the third stream (optionally of low priority)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05x; mkdir -p $OUT
run() { echo "$1 [$2]: $(env $2 timeout 120 python scripts/prof_hem.py 5000000 1 3 $1 2>&1 | grep 'rep2 L1 kernels' | cut -c16-140)"; }
for r in 1 2; do
for shape in iso; do
  run $shape "GSR_X=0"
  run $shape "GSR_HEM_SH_EARLY=128"
  run $shape "GSR_HEM_SH_EARLY=256"
  run $shape "GSR_HEM_SH_EARLY=512"
  run $shape "GSR_HEM_SH_EARLY=1024"
  run $shape "GSR_HEM_SH_EARLY=2048"
  run $shape "GSR_HEM_SH_EARLY=512 GSR_HEM_AUX2_LOW=1"
  run $shape "GSR_HEM_SH_EARLY=2048 GSR_HEM_AUX2_LOW=1"
done; done | tee $OUT/sh_beside_select.txt
