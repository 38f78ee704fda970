#!/bin/bash
# One GPU session producing the artefacts of a round: tests, smoke, bench line, rocprof summaries.
# usage (through gpurun): bash scripts/gpu_round.sh rNN
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls $OUT
