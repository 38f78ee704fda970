#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_icp_gpu.py -x -q > $OUT/icp_tests.log 2>&1; echo "icp tests: exit $?"; tail -6 $OUT/icp_tests.log
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -k "asynchronous or pair_partition or row_lists or two_pass" > $OUT/hem_some.log 2>&1; echo "hem subset: exit $?"; tail -3 $OUT/hem_some.log
for T in 0 -1 0 -1; do GSR_ICP_TILE=$T python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_tile$T.json 2> $OUT/bench_tile$T.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_tile$T.json").read().strip().splitlines()[-1])
print("GSR_ICP_TILE=$T ms/step %.2f hem %.2f icp %.2f value %.4g" % (d["ms_per_step"], d["hem_s_per_step"]*1e3, d["icp_s_per_step"]*1e3, d["value"]), [(l["ns"], l["iterations"], round(l["ms_per_iteration"],4)) for l in d["icp_per_level"]])
PY
done
