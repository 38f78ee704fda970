#!/bin/bash
# rocPRIM scan configurations (workgroup size, items per thread): time of the scan kernels per 5 M level
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD; OUT=gpurun_out/r05ak; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base $(cd $ABS/variants && ls scan_*.so | sed 's/.so//'); do
  if [ "$v" = "base" ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$ABS/variants/$v.so; fi
  rm -rf /tmp/trs; GSR_HEM_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trs -- python3 $ABS/scripts/prof_hem.py 5000000 1 4 > /tmp/log.txt 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/trs/**/*kernel_stats.csv",recursive=True)[0]
tot=0; det=[]
for r in csv.DictReader(open(f)):
    if "rocprim" in r["Name"] and "scan" in r["Name"] and "radix" not in r["Name"]:
        tot+=float(r["TotalDurationNs"]); det.append((int(r["Calls"]), round(float(r["AverageNs"])/1e3,1)))
lvl=[l for l in open("/tmp/log.txt") if "rep3 L1 kernels" in l]
print("$v: scan kernels per level %.1f us" % (tot/4e3), sorted(det, key=lambda x:-x[1])[:6], lvl[0].split("level")[-1].strip() if lvl else "")
PY
done | tee $ABS/$OUT/scan_configs.txt
