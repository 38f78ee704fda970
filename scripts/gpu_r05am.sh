#!/bin/bash
# rocPRIM merge-sort configurations (the sorts of up to 256 k keys): small-level wall times
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05am; mkdir -p $OUT
for r in 1 2 3; do for v in base $(cd variants && ls ms_*.so | sed 's/.so//'); do
  if [ "$v" = "base" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/variants/$v.so; fi
  echo "$v: $(python scripts/small_levels.py 2>&1 | tail -1)"
done; done | tee $OUT/merge_sort_configs.txt
