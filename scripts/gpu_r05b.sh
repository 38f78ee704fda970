#!/bin/bash
# Round 5, session B: the asynchronous level.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -k "asynchronous or survivor_ring" > $OUT/async_tests.log 2>&1; echo "async tests: exit $?"; tail -15 $OUT/async_tests.log
timeout 1800 python -m pytest tests/test_hem_gpu.py tests/test_configs_gpu.py -x -q > $OUT/hem_tests.log 2>&1; echo "hem+configs tests: exit $?"; tail -5 $OUT/hem_tests.log
python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_async.json 2> $OUT/bench_async.err; tail -c 300 $OUT/bench_async.json; echo
GSR_HEM_ASYNC=0 python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_sync.json 2> $OUT/bench_sync.err; tail -c 300 $OUT/bench_sync.json; echo
python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_async2.json 2> $OUT/bench_async2.err; tail -c 300 $OUT/bench_async2.json; echo
ls $OUT
