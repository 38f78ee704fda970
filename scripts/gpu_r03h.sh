#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python bench.py --no-cpu-baseline --steps 3 > gpurun_out/r03h_bench.json 2>gpurun_out/r03h_bench.err; tail -3 gpurun_out/r03h_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1])
print("value %.3e ms/step %.2f icp/s %.0f hem_s %.4f icp_s %.4f iters %d"%(d["value"],d["ms_per_step"],d["icp_iters_per_sec"],d["hem_s_per_step"],d["icp_s_per_step"],d["icp_iterations_per_step"]))
print(d["icp_per_level"]); print(d["icp_result"])
r=d["roofline"]; print(r["kernel"], r["frac"], r["kernels"], r["level1"])
print(d.get("aniso_level"))
PY
python bench.py --no-cpu-baseline --steps 2 --workload aniso 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ANISO pair: value %.3e ms/step %.2f hem_s %.4f icp_s %.4f'%(d['value'],d['ms_per_step'],d['hem_s_per_step'],d['icp_s_per_step']), d['config']['level_sizes'], d['icp_result'], [(l['ns'],l['iterations']) for l in d['icp_per_level']])"
