#!/bin/bash
# k_partition with 512- / 256-thread workgroups (more workgroups per CU to overlap each other's barrier phases), stages of 8192 / 4096 pairs; no SH copy beside it
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05ag; mkdir -p $OUT
export GSR_HEM_SH_DIRECT=1
run() { echo "$1 [$2]: $(env $2 timeout 120 python scripts/prof_hem.py 5000000 1 3 iso 2>&1 | grep 'rep2 L1 kernels' | grep -oE "partition [0-9.]+|level [0-9.]+" | tr '\n' ' ')"; }
for r in 1 2; do
  run base "GSR_X=0"
  run base-4096 "GSR_HEM_PARTITION_STAGE=4096"
  run t512 "GSR_HIP_LIB=$PWD/variants/pt512.so"
  run t512-4096 "GSR_HIP_LIB=$PWD/variants/pt512.so GSR_HEM_PARTITION_STAGE=4096"
  run t256-4096 "GSR_HIP_LIB=$PWD/variants/pt256.so GSR_HEM_PARTITION_STAGE=4096"
done | tee $OUT/partition_threads.txt
