#!/bin/bash
# The full randomised sweeps (tests/stress_*.py), logs kept: usage (through gpurun): bash scripts/gpu_stress.sh r04
set -uo pipefail
R=${1:-r04}
OUT=gpurun_out/${R}_stress
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p $OUT
python tests/stress_parity.py 400 2026 > $OUT/parity.txt 2>&1; tail -1 $OUT/parity.txt
STRESS_PATHOLOGIES=1 python tests/stress_parity.py 200 77 > $OUT/parity_pathologies.txt 2>&1; tail -1 $OUT/parity_pathologies.txt
python tests/stress_tiny.py 2000 5 > $OUT/tiny.txt 2>&1; tail -1 $OUT/tiny.txt
python tests/stress_knobs.py 40 31 > $OUT/knobs.txt 2>&1; tail -1 $OUT/knobs.txt
python tests/stress_icp.py 300 7 > $OUT/icp.txt 2>&1; tail -1 $OUT/icp.txt
python tests/stress_partition.py 30 3 99 > $OUT/partition_w3.txt 2>&1; tail -1 $OUT/partition_w3.txt
python tests/stress_partition.py 20 4 199 > $OUT/partition_w4.txt 2>&1; tail -1 $OUT/partition_w4.txt
python tests/stress_partition.py 12 8 299 > $OUT/partition_w8.txt 2>&1; tail -1 $OUT/partition_w8.txt
