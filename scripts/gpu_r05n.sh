#!/bin/bash
# the whole GPU suite on the packed-sum M-step (four rounds of row loads in flight), then the bench line
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; echo "gpu tests: exit $?"; grep -E "passed|failed|error" $OUT/gpu_tests.log | tail -3
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/step %.2f hem %.2f icp %.2f value %.4g" % (d["ms_per_step"], d["hem_s_per_step"]*1e3, d["icp_s_per_step"]*1e3, d["value"]), "level1", r["level1"]["ms_level"], r["level1"]["ms_k_select"], r["level1"]["ms_k_mstep"], "aniso", d["aniso_level"]["ms_level"], "clustered", d["clustered_level"]["ms_level"])
PY
