#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD
mkdir -p gpurun_out/r03t
cd /tmp && export TMPDIR=/tmp
for v in "0 0" "1 0" "1 800000"; do
  set -- $v
  export GSR_ICP_BLOCK_SEARCH=$1 GSR_ICP_WIDE_BELOW=$2
  rm -rf $ABS/gpurun_out/r03t/tr
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/gpurun_out/r03t/tr -- python3 $ABS/scripts/prof_icp_small.py 185000 2 0.5 > $ABS/gpurun_out/r03t/log_$1_$2.txt 2>&1
  grep "rep1" $ABS/gpurun_out/r03t/log_$1_$2.txt
  python3 - <<PY
import csv,glob
f=glob.glob("$ABS/gpurun_out/r03t/tr/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "icp_accumulate_dev" in r["Name"] or "icp_step" in r["Name"]:
        print("  block=$1 coop_below=$2", r["Name"][:60], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2), "max", round(float(r["MaxNs"])/1e3,2))
PY
done
rm -rf $ABS/gpurun_out/r03t/tr
