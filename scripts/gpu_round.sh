#!/bin/bash
# One GPU session producing the artefacts of a round: tests, smoke, bench line, rocprof summaries.
# usage (through gpurun): bash scripts/gpu_round.sh rNN [skip-tests]
set -uo pipefail
R=${1:-r02}
OUT=gpurun_out/$R
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p $OUT
if [ "${2:-}" != "skip-tests" ]; then
  python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
fi
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; echo
ABS=$PWD
cd /tmp && export TMPDIR=/tmp
# kernel trace + stats of the SAME bench command (the profiler's program is python3 itself, nothing in between)
rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/prof_stats -- python3 $ABS/bench.py --no-cpu-baseline > $ABS/$OUT/bench_prof.json 2> $ABS/$OUT/bench_prof.err
# counters: separate passes, level 1 of a 5 M-splat cloud alone
P="python3 $ABS/scripts/prof_hem.py 5000000 1 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ABS/$OUT/pmc_fetch -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ABS/$OUT/pmc_write -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $ABS/$OUT/pmc_l2 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $ABS/$OUT/pmc_sq -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d $ABS/$OUT/pmc_sq2 -- $P > /dev/null 2>&1
# the same FETCH/WRITE pass on the anisotropic (surfel-shaped) cloud: the second regime of bench.py's aniso_level
PA="python3 $ABS/scripts/prof_hem.py 5000000 1 2 aniso"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ABS/$OUT/pmc_aniso_fetch -- $PA > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ABS/$OUT/pmc_aniso_write -- $PA > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $ABS/$OUT/pmc_aniso_l2 -- $PA > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $ABS/$OUT/pmc_aniso_sq -- $PA > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d $ABS/$OUT/pmc_aniso_sq2 -- $PA > /dev/null 2>&1
# ICP: the coarse-to-fine schedule of the bench on the levels of a 5 M pair (scripts/prof_icp.py), counters in separate passes
PI="python3 $ABS/scripts/prof_icp.py 5000000 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/icp_stats -- $PI > $ABS/$OUT/prof_icp.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ABS/$OUT/pmc_icp_fetch -- $PI > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ABS/$OUT/pmc_icp_write -- $PI > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $ABS/$OUT/pmc_icp_l2 -- $PI > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $ABS/$OUT/pmc_icp_sq -- $PI > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d $ABS/$OUT/pmc_icp_sq2 -- $PI > /dev/null 2>&1
cd $ABS
# BASELINE configs[1] (a 1 M-splat pair) as its own line
python bench.py --splats 1000000 --no-cpu-baseline --no-aniso > $OUT/bench_c2.json 2> $OUT/bench_c2.err; tail -c 300 $OUT/bench_c2.json; echo
ls $OUT
