"""Which synthetic pairs can the coarse-to-fine ICP of bench.py register?  SURVEY.md 8(d) asks for a 5 degree / 0.05 h pair; at the
bench's correspondence distances [0.5, 0.3, 0.2, 0.1] that pair starts 0.5 - 1.5 scene units apart, more than max_corr and several
point spacings of the coarsest level.  This sweep registers pairs of growing misalignment (HEM levels recomputed per pair) and
prints error against the ground truth and the iterations used per level: the hardest pair that converges becomes the bench pair."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gaussiansplattingregistration_amd import hem, icp as icp_mod, synth
from gaussiansplattingregistration_amd.utils import local_registration_util as lru

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev = torch.device("cuda", 0)
tgt = synth.make_cloud_torch(n, seed=100, device=dev)
ctxs = {"hem": hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS), "icp": icp_mod.IcpContext(device=0)}
ctxs["hem"].set_rng("glibc", 1, 0)
tl, _ = bench.hem_levels(ctxs["hem"], tgt)
for ang, sh in ((1.0, 0.004), (1.5, 0.006), (2.0, 0.008), (2.0, 0.012), (3.0, 0.012), (3.0, 0.02), (4.0, 0.02), (5.0, 0.03), (5.0, 0.05)):
    T_gt = synth.rigid_transform(ang, (1, 1, 1), sh * tgt["h"] * np.array([1.0, -1.0, 0.5]))
    src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
    gen = torch.Generator(device=dev).manual_seed(7)
    src["xyz"] = src["xyz"] + torch.randn(src["xyz"].shape, device=dev, generator=gen) * 0.002
    src = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in src.items()}
    sl, _ = bench.hem_levels(ctxs["hem"], src, borrow=False)
    out = bench.coarse_to_fine(lru, ctxs["icp"], sl, tl, 0)
    err = float(np.linalg.norm(out["T"] - T_gt))
    print(f"angle {ang} deg shift {sh} h: |T - T_gt|_F {err:.2e} fitness {out['fitness']:.4f} rmse {out['rmse']:.5f} iterations {[l['iterations'] for l in out['levels']]}", flush=True)
