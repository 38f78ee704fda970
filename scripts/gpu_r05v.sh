#!/bin/bash
# where k_partition's time goes: variant libraries that leave the kernel after its 1st .. 4th phase (wrong sums, timing only); no SH copy beside it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05v; mkdir -p $OUT
export GSR_HEM_SH_DIRECT=1
for r in 1 2; do
for lib in - pskip1 pskip2 pskip3 pskip4; do
  if [ "$lib" = "-" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$lib.so; fi
  echo "$lib: $(timeout 120 python scripts/prof_hem.py 5000000 1 3 iso 2>&1 | grep 'rep2 L1 kernels' | cut -c1-160)"
done; done | tee $OUT/partition_phases.txt
