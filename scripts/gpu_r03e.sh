#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for f in 1 0 1 0; do
GSR_ICP_FUSED_STEP=$f python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused=$f: ms/step %.2f icp_s %.4f'%(d['ms_per_step'],d['icp_s_per_step']), [(l['ns'],l['iterations'],round(l['ms_per_iteration']*1e3,1)) for l in d['icp_per_level']])"
done
