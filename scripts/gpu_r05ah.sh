#!/bin/bash
# rocPRIM Onesweep configurations (workgroup size, keys per thread) for the level's two sorts: time of the rocprim kernels per 5 M level
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD; OUT=gpurun_out/r05ah; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base sort_1024_8_10 sort_1024_4_10 sort_1024_6_10 sort_1024_10_10 sort_1024_12_10 sort_1024_8_8 sort_1024_8_9 sort_1024_8_11; do
  if [ "$v" = "base" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$ABS/variants/$v.so; fi
  rm -rf /tmp/trs; GSR_HEM_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trs -- python3 $ABS/scripts/prof_hem.py 5000000 1 4 > /tmp/log.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/trs/**/*kernel_stats.csv",recursive=True)[0]
tot=0; det=[]
for r in csv.DictReader(open(f)):
    if "rocprim" in r["Name"] and ("radix" in r["Name"] or "onesweep" in r["Name"] or "histogram" in r["Name"]):
        tot+=float(r["TotalDurationNs"]); det.append((int(r["Calls"]), round(float(r["AverageNs"])/1e3,1)))
lvl=[l for l in open("/tmp/log.txt") if "rep3 L1 kernels" in l]
print("$v: radix-sort kernels per level %.1f us" % (tot/4e3), det, lvl[0].split("level")[-1].strip() if lvl else "")
PY
done | tee $ABS/$OUT/sort_configs2.txt
