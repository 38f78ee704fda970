#!/bin/bash
# M-step with packed cross-row sums: the HEM tests, then the bench line and the level ladder
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05l; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hem_gpu.py -x -q > $OUT/hem_tests.log 2>&1; echo "hem tests: exit $?"; tail -5 $OUT/hem_tests.log
for i in 1 2; do python bench.py --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/bench$i.err; python - <<PY
import json
d=json.loads(open("$OUT/bench$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/step %.2f hem %.2f icp %.2f value %.4g" % (d["ms_per_step"], d["hem_s_per_step"]*1e3, d["icp_s_per_step"]*1e3, d["value"]), "level1", r["level1"]["ms_level"], r["level1"]["ms_k_select"], r["level1"]["ms_k_mstep"], "aniso", d.get("aniso_level"), "clustered", d.get("clustered_level"))
PY
done
