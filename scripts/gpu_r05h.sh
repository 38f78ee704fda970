#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05h; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/full_gpu_tests.log 2>&1; echo "full gpu tests: exit $?"; grep -E "passed|failed" $OUT/full_gpu_tests.log | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-aniso > $OUT/bench$i.json 2> $OUT/bench$i.err; python - <<PY
import json
d=json.loads(open("$OUT/bench$i.json").read().strip().splitlines()[-1])
ph=d["hem_phase_ms_per_step"]
print("ms/step %.2f hem %.2f icp %.2f value %.4g | L1 %.2f ksel %.2f kmst %.2f grid %.2f sel %.2f sum %.2f mst %.2f" % (d["ms_per_step"], d["hem_s_per_step"]*1e3, d["icp_s_per_step"]*1e3, d["value"], d["roofline"]["level1"]["ms_level"], ph["ms_k_select"], ph["ms_k_mstep"], ph.get("ms_grid",0), ph.get("ms_select",0), ph.get("ms_sumlw",0), ph.get("ms_mstep",0)))
PY
done
python scripts/level_ladder.py iso 7 2>&1 | grep -v amdgpu.ids | tee $OUT/ladder_iso.txt
python scripts/level_ladder.py aniso 7 2>&1 | grep -v amdgpu.ids | tee $OUT/ladder_aniso.txt
