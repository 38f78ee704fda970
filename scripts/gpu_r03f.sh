#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r03f; mkdir -p $OUT
python -m pytest tests/test_planes_gpu.py tests/test_icp_gpu.py tests/test_distributed_gpu.py tests/test_ply_io.py -x -q -m gpu 2>&1 | tail -8
for f in 1 0 1 0; do
GSR_ICP_FUSED_STEP=$f python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused=$f: ms/step %.2f icp_s %.4f'%(d['ms_per_step'],d['icp_s_per_step']), [(l['ns'],l['iterations'],round(l['ms_per_iteration']*1e3,1)) for l in d['icp_per_level']])"
done
ABS=$PWD; cd /tmp; export TMPDIR=/tmp
for shape in iso aniso; do
rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/trace_$shape -- python3 $ABS/scripts/prof_hem.py 5000000 1 4 $shape > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$ABS/$OUT/trace_$shape/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
print("== $shape kernel stats (4 reps of one 5M level)")
for r in rows[:28]:
    print("%-70s calls %4s avg %9.1f us  tot %8.2f ms"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
