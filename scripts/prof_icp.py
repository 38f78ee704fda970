"""Where does an ICP call spend its time?  (5M x 5M, point-to-plane)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingregistration_amd import icp, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
tgt = synth.make_cloud_torch(n, seed=100)
T_gt = synth.rigid_transform(1.0, (1, 1, 1), 0.004 * tgt["h"] * np.array([1.0, -1.0, 0.5]))
src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
src["xyz"] = (src["xyz"] + torch.randn_like(src["xyz"]) * 0.002).contiguous()
def t(f, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
c = icp.IcpContext()
for rep in range(3):
    nrm, tn = t(icp.normals_from_cov, tgt["cov6"])
    _, tt = t(c.set_target, tgt["xyz"], nrm, 0.1)
    _, ts = t(c.set_source, src["xyz"])
    r, tr = t(c.register, np.eye(4), 1, 0, 0.0, 1e-6, 1e-6, 10)
    print(f"rep{rep}: normals {tn:.2f} ms, set_target {tt:.2f}, set_source {ts:.2f}, register {tr:.2f} ms ({r['iterations']} iters), timing {c.timing()}")
