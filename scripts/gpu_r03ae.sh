#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD
mkdir -p gpurun_out/r03ae
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export GSR_ICP_XCD=$v
  rm -rf $ABS/gpurun_out/r03ae/tr
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/gpurun_out/r03ae/tr -- python3 $ABS/scripts/prof_icp.py 5000000 2 > $ABS/gpurun_out/r03ae/log_$v.txt 2>&1
  grep "rep1" $ABS/gpurun_out/r03ae/log_$v.txt
  python3 - <<PY
import csv,glob
f=glob.glob("$ABS/gpurun_out/r03ae/tr/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "icp_nn" in r["Kernel_Name"] or "icp_accumulate" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
big=[(r["Kernel_Name"][:45], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]) for r in rows]
# last 12 launches with duration > 60us
sel=[b for b in big if b[1]>60][-12:]
for b in sel: print("  xcd=$v", b[0], round(b[1],1), "us grid", b[2])
PY
done
rm -rf $ABS/gpurun_out/r03ae/tr
