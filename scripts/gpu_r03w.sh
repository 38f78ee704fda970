#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03w
export GSR_ICP_WIDE_BELOW=0
for v in 512 768 1024 1536 2304 4096; do
  GSR_ICP_BLOCKS=$v python bench.py --no-cpu-baseline --no-aniso --steps 3 > gpurun_out/r03w/b_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03w/b_$v.json").read().strip().splitlines()[-1])
print("blocks $v", "icp_s", round(d["icp_s_per_step"]*1e3,3), [ (l["ns"], l["iterations"], round(l["ms_per_iteration"]*1e3,1)) for l in d["icp_per_level"]])
PY
done
