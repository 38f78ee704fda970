#!/bin/bash
# Round 5, session A: the ring fix (test on the old and the new library), configs[3] / configs[4] at full size on the one GPU, a bench line.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05a
mkdir -p $OUT
export GSR_MOCK_RCCL_SLOT_MB=1024
# 1. the survivor-ring regression: must FAIL on round 4's library and pass on this one
GSR_HIP_LIB=$PWD/variants/r04.so timeout 600 python -m pytest tests/test_hem_gpu.py -x -q -k "survivor_ring" > $OUT/ring_old.log 2>&1; echo "ring test on r04 library: exit $?"; tail -3 $OUT/ring_old.log
timeout 900 python -m pytest tests/test_hem_gpu.py -x -q -k "survivor_ring or parents_per_selection" > $OUT/ring_new.log 2>&1; echo "ring test on new library: exit $?"; tail -3 $OUT/ring_new.log
# 2. configs[4]: 40 M over 8 ranks on this GPU (mock RCCL), bit-compared with ONE context's 40 M levels
timeout 2400 python scripts/fullsize_modes.py c5 --splats 40000000 --world 8 --out $OUT/r05_c5_40m_8ranks.json > $OUT/c5.log 2>&1; echo "c5 full size: exit $?"; tail -2 $OUT/c5.log
# 3. configs[3]: 2 x 5 M over 2 ranks
timeout 1200 python scripts/fullsize_modes.py c4 --splats 5000000 --out $OUT/r05_c4_2x5m.json > $OUT/c4.log 2>&1; echo "c4 full size: exit $?"; tail -2 $OUT/c4.log
# 4. the -m gpu form of both (1 M per rank)
timeout 900 python -m pytest tests/test_fullsize_modes_gpu.py -x -q > $OUT/test_fullsize.log 2>&1; echo "fullsize tests: exit $?"; tail -3 $OUT/test_fullsize.log
# 5. bench.py --mode c5 at full size, 8 ranks on device 0 through the mock (timing of the whole step incl. ICP; no xGMI involved)
MOCK=$PWD/tests/mock_rccl/libmock_rccl.so
GSR_BENCH_SAME_DEVICE=1 GSR_DIST_BACKEND=gloo GSR_COMM_TRANSPORT=rccl GSR_RCCL_LIB=$MOCK timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 \
  --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 --mode c5 --splats 40000000 --target-splats 5000000 --steps 1 --warmup 1 \
  > $OUT/bench_c5_40m.json 2> $OUT/bench_c5_40m.err; echo "bench c5 40M: exit $?"; tail -c 400 $OUT/bench_c5_40m.json; echo
GSR_BENCH_SAME_DEVICE=1 GSR_DIST_BACKEND=gloo GSR_COMM_TRANSPORT=rccl GSR_RCCL_LIB=$MOCK timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 \
  --master-addr 127.0.0.1 --master-port 29657 bench.py --gpus 2 --mode c4 --splats 5000000 --steps 2 --warmup 1 \
  > $OUT/bench_c4_5m.json 2> $OUT/bench_c4_5m.err; echo "bench c4 5M: exit $?"; tail -c 400 $OUT/bench_c4_5m.json; echo
# 6. the default bench line (this round's starting point) and the A/B of the ring fix on the 5 M level
python bench.py --no-cpu-baseline > $OUT/bench_start.json 2> $OUT/bench_start.err; tail -c 300 $OUT/bench_start.json; echo
GSR_HIP_LIB=$PWD/variants/r04.so python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_r04lib.json 2> $OUT/bench_r04lib.err; tail -c 300 $OUT/bench_r04lib.json; echo
python bench.py --no-cpu-baseline --no-aniso > $OUT/bench_start2.json 2> $OUT/bench_start2.err; tail -c 300 $OUT/bench_start2.json; echo
ls $OUT
