#!/bin/bash
# the bench's own cascade: M-step at four waves per SIMD (four rounds of row loads) against five (two / three rounds)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05aa; mkdir -p $OUT
for round in 1 2 3; do for v in u4w0 u2w5 u3w5; do echo "round $round $v: $(GSR_HIP_LIB=$PWD/variants/$v.so python bench.py --no-cpu-baseline --no-aniso --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print(round(d['ms_per_step'],3), 'hem', round(d['hem_s_per_step']*1e3,3), 'icp', round(d['icp_s_per_step']*1e3,3), 'mstep/step', round(k['k_mstep']['total_ms_per_step'],3), 'select/step', round(k['k_select']['total_ms_per_step'],3))")"; done; done | tee $OUT/ab_bench_mstep_waves.txt
