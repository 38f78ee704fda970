#!/bin/bash
# ICP, distributed and bench-mode tests, then one bench step: ICP per-level iteration times (through gpurun)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/icp_check
timeout 1200 python -m pytest tests/test_icp_gpu.py tests/test_distributed_gpu.py tests/test_bench_modes_gpu.py -x -q -m gpu > gpurun_out/icp_check/t.log 2>&1; grep -E "passed|failed" gpurun_out/icp_check/t.log
python bench.py --no-cpu-baseline --no-aniso --steps 3 > gpurun_out/icp_check/b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open("gpurun_out/icp_check/b.json").read().strip().splitlines()[-1])
print("icp_s", round(d["icp_s_per_step"]*1e3,3), [ (l["ns"], l["iterations"], round(l["ms_per_iteration"]*1e3,1)) for l in d["icp_per_level"]], "ms/step", round(d["ms_per_step"],2), "hem", round(d["hem_s_per_step"]*1e3,2), "value", d["value"])
PY
