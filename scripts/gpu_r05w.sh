#!/bin/bash
# what a compact pair layout is worth to the kernels behind the selection: the COUNT + FILL fallback writes compact CSR pairs
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05w; mkdir -p $OUT
for r in 1 2; do
for mode in "sparse" "compact"; do
  if [ "$mode" = "compact" ]; then export GSR_HEM_SPARSE_GB=0; else unset GSR_HEM_SPARSE_GB; fi
  for shd in 0 1; do
    export GSR_HEM_SH_DIRECT=$shd
    echo "$mode sh_direct=$shd: $(timeout 120 python scripts/prof_hem.py 5000000 1 3 iso 2>&1 | grep 'rep2 L1 kernels' | cut -c1-170)"
  done
done; done | tee $OUT/compact_vs_sparse.txt
