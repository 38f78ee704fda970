#!/bin/bash
# per-kernel tables of ONE 5 M level (three repetitions) on the isotropic and the surfel cloud, and the surfel bench step under --stats
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD; OUT=gpurun_out/r05d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SHAPE in iso aniso; do
rm -rf $ABS/$OUT/tr
GSR_HEM_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/tr -- python3 $ABS/scripts/prof_hem.py 5000000 1 4 $SHAPE > $ABS/$OUT/log_$SHAPE.txt 2>&1
python3 - <<PY > $ABS/$OUT/level_kernels_$SHAPE.txt
import csv,glob
f=glob.glob("$ABS/$OUT/tr/**/*kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r["TotalDurationNs"]))
tot=0
for r in rows:
    if r["Name"].startswith("void at::") or "distribution" in r["Name"]: continue
    tot+=float(r["TotalDurationNs"])/4e3
    print(f'{r["Name"][:80]:80s} calls/level {int(r["Calls"])/4:5.1f}  avg us {float(r["AverageNs"])/1e3:8.1f}  per level us {float(r["TotalDurationNs"])/4e3:8.1f}')
print("sum per level us %.1f" % tot)
PY
rm -rf $ABS/$OUT/tr
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/prof_stats -- python3 $ABS/bench.py --workload aniso --no-cpu-baseline > $ABS/$OUT/bench_aniso.json 2> $ABS/$OUT/bench_aniso.err
cd $ABS
python3 scripts/summarize_profiles.py $OUT $OUT/r05d_aniso > /dev/null 2>&1
rm -rf $OUT/prof_stats
grep "rep3 L1" $OUT/log_iso.txt | cut -c1-400; grep "rep3 L1" $OUT/log_aniso.txt | cut -c1-400
head -45 $OUT/level_kernels_aniso.txt
ls $OUT
