#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=gpurun_out/r05j; mkdir -p $OUT
export GSR_MOCK_RCCL_SLOT_MB=1024
MOCK=$PWD/tests/mock_rccl/libmock_rccl.so
GSR_BENCH_SAME_DEVICE=1 GSR_DIST_BACKEND=gloo GSR_COMM_TRANSPORT=rccl GSR_RCCL_LIB=$MOCK timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 \
  --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 --mode c5 --splats 40000000 --target-splats 5000000 --steps 1 --warmup 1 \
  > $OUT/bench_c5_40m.json 2> $OUT/bench_c5_40m.err; echo "bench c5 40M: exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_c5_40m.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","hem_s_per_step","icp_s_per_step","icp_iterations_per_step","transport")}, d["icp_result"], d["config"]["level_sizes"], d["config"]["pair"])
PY
# one context, 40 M, warm: the level times of the single-GPU path at this size
python - <<PY 2>&1 | grep -v amdgpu | tee $OUT/single_40m_warm.txt
import sys, time, torch
sys.path.insert(0, ".")
import bench
from gaussiansplattingregistration_amd import hem, synth
dev = torch.device("cuda", 0)
parts = [synth.make_block_cloud_torch(40_000_000, r, 8, seed=100, device=dev)[0] for r in range(8)]
c = {k: torch.cat([p[k] for p in parts]).contiguous() for k in ("xyz", "color", "opacity", "cov6", "sh")}
del parts
m = hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS)
for rep in range(3):
    m.set_rng("glibc", 1, 0)
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)
    for l in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        m.run_level()
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        st = m.stats()
        print("rep %d L%d: n_in %d wall %.2f ms events %.2f schedule %d round trips %d" % (rep, l + 1, st["n_in"], dt * 1e3, st["ms_level"], st["schedule"], st["round_trips"]), flush=True)
PY
