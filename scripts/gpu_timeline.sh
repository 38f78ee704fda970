#!/bin/bash
# kernel timeline (durations and idle gaps) of one HEM level: bash scripts/gpu_timeline.sh N   (through gpurun)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
ABS=$PWD; N=${1:-200000}
mkdir -p gpurun_out/timeline
cd /tmp && export TMPDIR=/tmp
rm -rf $ABS/gpurun_out/timeline/tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ABS/gpurun_out/timeline/tr -- python3 $ABS/scripts/prof_hem.py $N 1 3 > $ABS/gpurun_out/timeline/log.txt 2>&1
cd $ABS
python scripts/trace_timeline.py gpurun_out/timeline/tr k_prep > gpurun_out/timeline/timeline_$N.txt
rm -rf gpurun_out/timeline/tr
grep "kernels" gpurun_out/timeline/log.txt | tail -1
tail -40 gpurun_out/timeline/timeline_$N.txt | grep -E "window|before" | head -40
