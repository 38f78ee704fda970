#!/bin/bash
# counters of the 5 M isotropic level alone (the passes of gpu_round.sh), after the packed sums of the M-step
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
R=r05o; OUT=gpurun_out/$R; mkdir -p $OUT
ABS=$PWD
cd /tmp && export TMPDIR=/tmp
P="python3 $ABS/scripts/prof_hem.py 5000000 1 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $ABS/$OUT/prof_stats -- $P > $ABS/$OUT/prof_hem.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ABS/$OUT/pmc_fetch -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ABS/$OUT/pmc_write -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d $ABS/$OUT/pmc_l2 -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $ABS/$OUT/pmc_sq -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR --output-format csv -d $ABS/$OUT/pmc_sq2 -- $P > /dev/null 2>&1
cd $ABS; ls $OUT; tail -5 $OUT/prof_hem.log
