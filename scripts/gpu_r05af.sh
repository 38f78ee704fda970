#!/bin/bash
# configs[3] / configs[4] at full size again, on the round's final kernels
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05af; mkdir -p $OUT
export GSR_MOCK_RCCL_SLOT_MB=1024
timeout 2400 python scripts/fullsize_modes.py c5 --splats 40000000 --world 8 --out $OUT/r05_c5_40m_8ranks.json > $OUT/c5.log 2>&1; echo "c5 full size: exit $?"; tail -1 $OUT/c5.log | cut -c1-400
timeout 1200 python scripts/fullsize_modes.py c4 --splats 5000000 --out $OUT/r05_c4_2x5m.json > $OUT/c4.log 2>&1; echo "c4 full size: exit $?"; tail -1 $OUT/c4.log | cut -c1-400
