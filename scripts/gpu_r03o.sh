#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD
mkdir -p gpurun_out/r03o
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ABS/gpurun_out/r03o/t5m -- python3 $ABS/scripts/prof_hem.py 5000000 1 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $ABS/gpurun_out/r03o/t200k -- python3 $ABS/scripts/prof_hem.py 200000 1 3 > $ABS/gpurun_out/r03o/p200k.log 2>&1
cd $ABS
python scripts/trace_timeline.py gpurun_out/r03o/t5m k_prep > gpurun_out/r03o/timeline_5m.txt
python scripts/trace_timeline.py gpurun_out/r03o/t200k k_prep > gpurun_out/r03o/timeline_200k.txt
rm -rf gpurun_out/r03o/t5m gpurun_out/r03o/t200k
tail -25 gpurun_out/r03o/timeline_5m.txt
grep "kernels" gpurun_out/r03o/p200k.log
