#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03p
timeout 1200 python -m pytest tests/test_icp_gpu.py -x -q -m gpu > gpurun_out/r03p/t.log 2>&1; grep -E "passed|failed" gpurun_out/r03p/t.log
for v in 0 300000 800000; do
  GSR_ICP_WIDE_BELOW=$v python bench.py --no-cpu-baseline --no-aniso --steps 3 > gpurun_out/r03p/b_$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03p/b_$v.json").read().strip().splitlines()[-1])
print("wide_below $v", "icp_s", round(d["icp_s_per_step"]*1e3,3), [ (l["ns"], l["iterations"], round(l["ms_per_iteration"]*1e3,1)) for l in d["icp_per_level"]], "ms/step", round(d["ms_per_step"],2), "hem", round(d["hem_s_per_step"]*1e3,2))
PY
done
