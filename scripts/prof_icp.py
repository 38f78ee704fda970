"""Where the coarse-to-fine ICP of the bench spends its time: per level, the pieces of one registration_icp call -- on bench.py's
OWN pair (5 degrees about (1,1,1)/sqrt(3), 0.05 h apart, + 0.002 jitter: 50 / 29 / 4 / 2 iterations).  A marker kernel
(gsr_debug_logf on one float: `k_debug_logf` in the trace) is launched in front of every level, so that the counter summaries can be
split per level (scripts/summarize_profiles.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingregistration_amd import hem, icp as icp_mod, synth
from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
from gaussiansplattingregistration_amd.utils import local_registration_util as lru

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
sync = torch.cuda.synchronize
import bench
from gaussiansplattingregistration_amd import _lib
tgt = synth.make_cloud_torch(n, seed=100)
T_gt = synth.rigid_transform(bench.PAIR_ANGLE_DEG, (1, 1, 1), bench.PAIR_SHIFT_H * tgt["h"] * np.array([1.0, -1.0, 0.5]))
src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
gen = torch.Generator(device=dev).manual_seed(7)
src["xyz"] = src["xyz"] + torch.randn(src["xyz"].shape, device=dev, generator=gen) * 0.002
src = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in src.items()}
_L = _lib.load()
_one = np.ones(1, np.float32); _out = np.zeros(1, np.float32)
def marker():
    _L.gsr_debug_logf(_one.ctypes.data, 1, _out.ctypes.data, 0)
m = hem.HemMixture()
def levels(c):
    lv = [PointCloud(xyz32=c["xyz"], cov6=c["cov6"])]
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
    for _ in range(3):
        m.run_level(); d = m.get_level(as_torch=True); lv.append(PointCloud(xyz32=d["xyz"], cov6=d["cov6"]))
    return lv
m.set_rng("glibc", 1, 0)
S, Tg = levels(src), levels(tgt)
ctx = icp_mod.IcpContext(device=0)
est = lru.get_estimation(lru.LocalRegistrationType.ICP_Point_To_Plane, lru.RobustLoss(0))
ITER = bench.ITER_VALUES; MC = bench.MAX_CORR
def tm(f):
    sync(); t = time.perf_counter(); r = f(); sync(); return (time.perf_counter() - t) * 1e3, r
for rep in range(reps):
    T = np.eye(4); tot = 0.0
    for k in range(4):
        s, t = S[-(k + 1)], Tg[-(k + 1)]
        marker()
        t.normals = None
        a, _ = tm(lambda: t.estimate_normals())
        b, _ = tm(lambda: ctx.set_target(t.xyz32, t.normals, MC[k]))
        c, _ = tm(lambda: ctx.set_source(s.xyz32))
        d0, _ = tm(lambda: (ctx.set_allreduce_device(None, 0), ctx.set_allreduce(None, 0)))
        d, r = tm(lambda: ctx.register(T, est.kind, 0, 0.0, 1e-6, 1e-6, ITER[k]))
        tmg = ctx.timing()
        T = r["transformation"]; tot += a + b + c + d + d0
        print(f"rep{rep} ns={len(s):8d} normals {a:.3f} set_target {b:.3f} (build {tmg['ms_build']:.3f}) set_source {c:.3f} callbacks {d0:.3f} "
              f"register {d:.3f} (iters {r['iterations']}, kernels {tmg['ms_iters']:.3f})", flush=True)
    print(f"rep{rep} total {tot:.3f} ms  |T - T_gt|_F {np.linalg.norm(T - T_gt):.2e}", flush=True)
