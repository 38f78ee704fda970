#!/bin/bash
# k_mstep at 16 / 12 / 8 waves per CU (LDS padding): how much does it live on occupancy?
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05y; mkdir -p $OUT
bash scripts/ab_libs.sh iso - occ12 occ8 2>&1 | tee $OUT/mstep_occupancy.txt
