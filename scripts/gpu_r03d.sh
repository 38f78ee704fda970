#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests/test_icp_gpu.py tests/test_distributed_gpu.py tests/test_sparse_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -15
for ov in 2 1 0 2 0; do
echo "== iso overlap=$ov"; GSR_HEM_SH_OVERLAP=$ov python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1 | cut -c1-24,180-420
done
python bench.py --no-cpu-baseline --steps 3 > gpurun_out/r03d_bench.json 2>gpurun_out/r03d_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d_bench.json').read().strip().splitlines()[-1])
print("value %.3e ms/step %.2f icp/s %.0f hem_s %.4f icp_s %.4f"%(d["value"],d["ms_per_step"],d["icp_iters_per_sec"],d["hem_s_per_step"],d["icp_s_per_step"]))
print(d["icp_per_level"])
print(d["hem_phase_ms_per_step"])
PY
GSR_ICP_FUSED_STEP=0 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('unfused: icp_s %.4f'%d['icp_s_per_step'], d['icp_per_level'])"
