#!/bin/bash
# Round 5, session E: SH rows read in place when the pair count says so (k_gather_sh decides on the device), fused orphan kernel: tests + A/B
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -k "sh_rows_read or asynchronous or small_parent or heavy_parent or pair_partition" > $OUT/tests_direct.log 2>&1; echo "direct tests: exit $?"; tail -4 $OUT/tests_direct.log
for SHAPE in iso aniso clustered; do for D in 0 1 auto 0 1 auto; do
  if [ $D = auto ]; then unset GSR_HEM_SH_DIRECT; else export GSR_HEM_SH_DIRECT=$D; fi
  echo "== $SHAPE SH_DIRECT=$D"; GSR_HEM_TIMING=1 python scripts/prof_hem.py 5000000 1 5 $SHAPE 2>&1 | grep "kernels" | tail -2
done; done 2>&1 | tee $OUT/ab_sh_direct.txt
unset GSR_HEM_SH_DIRECT
for D in 0 auto; do if [ $D = auto ]; then unset GSR_HEM_SH_DIRECT; else export GSR_HEM_SH_DIRECT=$D; fi; python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_direct$D.json 2> $OUT/bench_direct$D.err; tail -c 200 $OUT/bench_direct$D.json; echo; done
unset GSR_HEM_SH_DIRECT
timeout 1800 python -m pytest tests/test_hem_gpu.py tests/test_configs_gpu.py tests/test_stress_gpu.py tests/test_distributed_gpu.py -x -q > $OUT/hem_tests.log 2>&1; echo "hem+configs+stress+distributed tests: exit $?"; tail -3 $OUT/hem_tests.log
