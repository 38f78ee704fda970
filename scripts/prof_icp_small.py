"""One ICP level alone (n source / n target points, point-to-plane): a short target for rocprofv3 --kernel-trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingregistration_amd import icp as icp_mod, synth
from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
from gaussiansplattingregistration_amd.utils import local_registration_util as lru

n = int(sys.argv[1]) if len(sys.argv) > 1 else 185_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
src = synth.make_cloud_torch(n, seed=100)
tgt = synth.apply_rigid_torch(src, synth.rigid_transform(1.0, translation=(0.004, -0.002, 0.003)))
s = PointCloud(xyz32=src["xyz"], cov6=src["cov6"]); t = PointCloud(xyz32=tgt["xyz"], cov6=tgt["cov6"])
t.estimate_normals()
ctx = icp_mod.IcpContext(device=0)
est = lru.get_estimation(lru.LocalRegistrationType.ICP_Point_To_Plane, lru.RobustLoss(0))
for rep in range(reps):
    ctx.set_target(t.xyz32, t.normals, mc); ctx.set_source(s.xyz32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.register(np.eye(4), est.kind, 0, 0.0, 1e-6, 1e-6, 50)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    tm = ctx.timing()
    print(f"rep{rep} n={n} iters {r['iterations']} wall {dt:.3f} ms kernels {tm['ms_iters']:.3f} ms -> {tm['ms_iters'] / (r['iterations'] + 1) * 1e3:.1f} us per evaluation", flush=True)
