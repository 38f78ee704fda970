#!/bin/bash
# round 3, first session: baselines (iso + aniso), SH-gather overlap A/B, FETCH_SIZE calibration
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r03a; mkdir -p $OUT
for i in 1 2; do
  echo "== iso overlap=1"; GSR_HEM_SH_OVERLAP=1 python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1
  echo "== iso overlap=0"; GSR_HEM_SH_OVERLAP=0 python scripts/prof_hem.py 5000000 3 2 iso 2>&1 | grep rep1
done
echo "== aniso"; python scripts/prof_hem.py 5000000 3 2 aniso 2>&1 | grep rep1
echo "== aniso 1M"; python scripts/prof_hem.py 1000000 3 2 aniso 2>&1 | grep rep1
ABS=$PWD; cd /tmp; export TMPDIR=/tmp
$ABS/scripts/micro/fetch_calib > $ABS/$OUT/fetch_calib.txt 2>&1; cat $ABS/$OUT/fetch_calib.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ABS/$OUT/pmc_calib -- $ABS/scripts/micro/fetch_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $ABS/$OUT/pmc_calib2 -- $ABS/scripts/micro/fetch_calib > /dev/null 2>&1
cd $ABS
python - <<'PY'
import csv, glob, collections
for d in ("pmc_calib","pmc_calib2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/r03a/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(d, k, {c:(vals[-1]) for c,vals in v.items()})
PY
