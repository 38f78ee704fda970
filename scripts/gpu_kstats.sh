#!/bin/bash
# per-kernel averages of one HEM level: bash scripts/gpu_kstats.sh N [iso|aniso]   (through gpurun)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD; N=${1:-5000000}; SHAPE=${2:-iso}
mkdir -p gpurun_out/kstats
cd /tmp && export TMPDIR=/tmp
rm -rf $ABS/gpurun_out/kstats/tr
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/gpurun_out/kstats/tr -- python3 $ABS/scripts/prof_hem.py $N 1 3 $SHAPE > $ABS/gpurun_out/kstats/log.txt 2>&1
cd $ABS
grep "rep2 L1" gpurun_out/kstats/log.txt | cut -c1-600
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kstats/tr/**/*kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:70]:
    if r["Name"].startswith("void at::") or "rocclr" in r["Name"] or "distribution" in r["Name"]: continue
    print(f'{r["Name"][:72]:72s} calls/level {int(r["Calls"])/3:5.1f}  avg us {float(r["AverageNs"])/1e3:8.1f}  per level us {float(r["TotalDurationNs"])/3e3:8.1f}')
PY
rm -rf gpurun_out/kstats/tr
