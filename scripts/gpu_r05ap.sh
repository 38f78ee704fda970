#!/bin/bash
# components per grid cell (GSR_HEM_CELL_TARGET) again, on the round's final kernels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05ap; mkdir -p $OUT
run() { echo "$1 [$2]: $(env $2 timeout 120 python scripts/prof_hem.py 5000000 3 3 $1 2>&1 | grep 'rep2 L. kernels' | grep -oE " select [0-9.]+| mstep [0-9.]+|level [0-9.]+" | tr '\n' ' ')"; }
for r in 1 2; do for shape in iso aniso clustered; do
  for t in 10 13 16 20 26; do run $shape "GSR_HEM_CELL_TARGET=$t"; done
done; done | tee $OUT/cell_target.txt
