#!/bin/bash
# the M-step with 7 424 bytes of LDS per wave: four waves per SIMD (four rounds of row loads) against five (two / three rounds)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05z; mkdir -p $OUT
GSR_HIP_LIB=$PWD/variants/w5u3.so timeout 1500 python -m pytest tests/test_hem_gpu.py -x -q > $OUT/tests_w5u3.log 2>&1; echo "tests (w5u3): exit $?"; grep -E "passed|failed|error" $OUT/tests_w5u3.log | tail -3
for shape in iso aniso clustered; do bash scripts/ab_libs.sh $shape - w5u2 w5u3 2>&1 | tee -a $OUT/ab_mstep_waves2.txt; done
