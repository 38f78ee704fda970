#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05c; mkdir -p $OUT; ABS=$PWD
python scripts/level_ladder.py iso 7 2>&1 | grep -v amdgpu.ids | tee $OUT/ladder_iso.txt
cd /tmp && export TMPDIR=/tmp
for N in 200000 556000; do
rm -rf $ABS/$OUT/tr
GSR_HEM_TIMING=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ABS/$OUT/tr -- python3 $ABS/scripts/prof_hem.py $N 1 4 > $ABS/$OUT/log_$N.txt 2>&1
python3 $ABS/scripts/trace_timeline.py $ABS/$OUT/tr k_keys > $ABS/$OUT/timeline_async_$N.txt
rm -rf $ABS/$OUT/tr
done
cd $ABS
tail -60 $OUT/timeline_async_200000.txt
