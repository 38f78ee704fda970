#!/bin/bash
# A/B on one box: the committed library (variants/head.so) against the tree (join before the SH copy's fork, orphans in input order)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05t; mkdir -p $OUT
for shape in iso aniso clustered; do bash scripts/ab_libs.sh $shape - head 2>&1 | tee -a $OUT/ab.txt; done
