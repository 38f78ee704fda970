"""The reference's DEFAULT local registration (LocalRegistrationParams, registration_parameters.py:7-15: point-to-point, max_correspondence 5.0 scene
units, 30 iterations) on a pair of n splats: time per iteration.  max_correspondence = 5 spans the whole synthetic scene; what matters is that the
exact search stays a NEAREST-neighbour search (a few candidates per query) and does not degrade into scanning cells of max_corr / 8.
usage: python scripts/icp_default_params.py [n=1000000] [max_corr=5.0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingregistration_amd import icp as icp_mod, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mc = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
dev = torch.device("cuda", 0)
tgt = synth.make_cloud_torch(n, seed=100, device=dev)
T_gt = synth.rigid_transform(2.0, (1, 1, 1), 0.01 * tgt["h"] * np.array([1.0, -1.0, 0.5]))
src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
src["xyz"] = (src["xyz"] + torch.randn_like(src["xyz"]) * 0.002).contiguous()
with icp_mod.IcpContext(device=0) as c:
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = c.register_clouds(src["xyz"], tgt["xyz"], None, mc, np.eye(4), 0, 0, 0.0, 1e-6, 1e-6, 30)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        tm = c.timing()
        print(f"n {n} max_corr {mc}: {r['iterations']} iterations, {dt * 1e3:.2f} ms total, build {tm['ms_build']:.3f} ms, {tm['ms_iters'] / max(1, tm['iter_kernels']):.3f} ms per iteration, "
              f"fitness {r['fitness']:.4f} rmse {r['inlier_rmse']:.5f} |T - T_gt| {np.linalg.norm(r['transformation'] - T_gt):.2e}", flush=True)
