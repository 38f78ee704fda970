#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r03y
for n in 200000 556000 1670000; do python scripts/prof_hem.py $n 1 3 2>&1 | grep "rep2 L1 kernels" | sed -e "s/^/n=$n /"; done
python bench.py --no-cpu-baseline --no-aniso --steps 3 > gpurun_out/r03y/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/r03y/b.json").read().strip().splitlines()[-1])
print("icp_s", round(d["icp_s_per_step"]*1e3,3), [ (l["ns"], l["iterations"], round(l["ms_per_iteration"]*1e3,1), round(l["ms_target_index_build"]*1e3)) for l in d["icp_per_level"]], "ms/step", round(d["ms_per_step"],2), "hem", round(d["hem_s_per_step"]*1e3,2), "value", d["value"])
PY
