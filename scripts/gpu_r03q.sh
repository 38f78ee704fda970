#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03q
export GSR_ICP_WIDE_BELOW=0
for v in 0.7 1 1.4 2 3 4; do
  GSR_ICP_CELL_TARGET=$v python bench.py --no-cpu-baseline --no-aniso --steps 2 --warmup 1 > gpurun_out/r03q/b_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03q/b_$v.json").read().strip().splitlines()[-1])
print("cell_target $v", "icp_s", round(d["icp_s_per_step"]*1e3,3), [ (l["ns"], l["iterations"], round(l["ms_per_iteration"]*1e3,1), round(l["ms_target_index_build"]*1e3)) for l in d["icp_per_level"]])
PY
done
