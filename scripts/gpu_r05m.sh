#!/bin/bash
# rounds of SH row loads in flight, now that the packed sums freed 20 registers
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05m; mkdir -p $OUT
for shape in iso aniso; do bash scripts/ab_libs.sh $shape - mstepU2 mstepU4 mstepU5 2>&1 | tee -a $OUT/ab_mstep_u.txt; done
