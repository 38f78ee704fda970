#!/bin/bash
# Onesweep keys per thread in the ICP's target / source sorts (16 / 12 / 10 / 8)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05aj; mkdir -p $OUT
for r in 1 2 3; do for v in base icpms_1024_8 icpms_1024_4 icpms_512_8; do
  if [ "$v" = "base" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$v.so; fi
  echo "$v: $(python scripts/prof_icp.py 5000000 3 2>&1 | grep -E "^rep2 (ns|total)" | sed -E 's/callbacks [0-9.]+ //; s/normals [0-9.]+ //' | tr '\n' '|' | cut -c1-420)"
done; done | tee $OUT/icp_merge_cfg.txt
