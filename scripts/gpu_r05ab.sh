#!/bin/bash
# SH rows from the copy or from the level's own array, again, with the M-step at five waves per SIMD
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05ab; mkdir -p $OUT
run() { echo "$1 [$2]: $(env $2 timeout 120 python scripts/prof_hem.py 5000000 3 3 $1 2>&1 | grep 'rep2 L. kernels' | grep -oE " select [0-9.]+| mstep [0-9.]+|level [0-9.]+" | tr '\n' ' ')"; }
for r in 1 2 3; do
for shape in iso clustered; do
  run $shape "GSR_X=0"
  run $shape "GSR_HEM_SH_DIRECT=1"
  run $shape "GSR_HEM_SH_DIRECT=0"
done; done | tee $OUT/sh_direct_again.txt
for r in 1 2; do for v in "GSR_X=0" "GSR_HEM_SH_DIRECT=1"; do echo "bench [$v]: $(env $v python bench.py --no-cpu-baseline --no-aniso --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'hem', round(d['hem_s_per_step']*1e3,3), [round(l['ms_level'],3) for l in d['hem_levels_last_step']])")"; done; done | tee -a $OUT/sh_direct_again.txt
