set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r06s; mkdir -p $OUT
timeout 1500 python tests/stress_parity.py 400 606 > $OUT/stress_parity_400.txt 2>&1; tail -1 $OUT/stress_parity_400.txt
STRESS_PATHOLOGIES=1 timeout 1200 python tests/stress_parity.py 200 607 > $OUT/stress_parity_pathologies_200.txt 2>&1; tail -1 $OUT/stress_parity_pathologies_200.txt
timeout 900 python tests/stress_tiny.py 2000 608 > $OUT/stress_tiny_2000.txt 2>&1; tail -1 $OUT/stress_tiny_2000.txt
timeout 900 python tests/stress_knobs.py 40 609 > $OUT/stress_knobs_40.txt 2>&1; tail -1 $OUT/stress_knobs_40.txt
timeout 1500 python tests/stress_icp.py 300 610 > $OUT/stress_icp_300.txt 2>&1; tail -1 $OUT/stress_icp_300.txt
for w in 3 4 8; do timeout 900 python tests/stress_partition.py 20 $w 61$w > $OUT/stress_partition_w$w.txt 2>&1; tail -1 $OUT/stress_partition_w$w.txt; done
