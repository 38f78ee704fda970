#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD
mkdir -p gpurun_out/r03ab
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/gpurun_out/r03ab/tr -- python3 $ABS/scripts/prof_hem.py 5000000 1 3 > $ABS/gpurun_out/r03ab/log.txt 2>&1
cd $ABS
grep "rep2 L1 kernels" gpurun_out/r03ab/log.txt
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/r03ab/tr/**/*kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:24]:
    print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1))
PY
rm -rf gpurun_out/r03ab/tr
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
