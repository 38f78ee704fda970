"""First GPU bring-up: HEM levels + ICP against the oracle, with diagnostics."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussiansplattingregistration_amd import hem, icp, synth
from oracle import oracle as O

def cmp_levels(got, want, tag):
    ok = True
    for k in range(len(want)):
        ng, nw = got[k]["xyz"].shape[0], want[k]["xyz"].shape[0]
        line = f"{tag} level {k+1}: gpu n={ng} oracle n={nw}"
        if ng == nw:
            for name in ("xyz", "color", "cov6", "opacity", "sh"):
                a, b = got[k][name].astype(np.float64), want[k][name].astype(np.float64)
                if a.size == 0: continue
                err = np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-30)
                line += f" {name}:{err:.2e}"
                if not err < 1e-4: ok = False
        else:
            ok = False
        print(line, flush=True)
    return ok

print("torch", torch.__version__, torch.cuda.get_device_name(0), flush=True)
for n, h, L in ((400, 0.35, 2), (3000, 0.6, 3), (50000, 1.5, 3)):
    cloud = synth.make_cloud(n, seed=0, h=h)
    t = time.time(); want, wst = O.hem(cloud, L); to = time.time() - t
    t = time.time(); got, st = hem.create_mixture(cloud, L, device=0); tg = time.time() - t
    print(f"n={n}: oracle {to:.3f}s gpu(wall incl. alloc) {tg:.3f}s", flush=True)
    for s, w in zip(st, wst):
        print("   gpu ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items()})
        print("   orac", {k: w[k] for k in ("parents", "pairs", "orphans", "dropped", "candidates")})
    cmp_levels(got, want, f"n={n}")

# timing at larger sizes (GPU only)
for n in (200000, 1000000):
    cloud = synth.make_cloud(n, seed=0)
    dev = {k: torch.from_numpy(v).cuda() for k, v in cloud.items() if isinstance(v, np.ndarray)}
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        got, st = hem.create_mixture(dev, 3, device=0, as_torch=True)
        torch.cuda.synchronize(); tg = time.time() - t
        print(f"n={n} rep{rep}: 3 levels wall {tg*1e3:.1f} ms; sizes {[g['xyz'].shape[0] for g in got]}", flush=True)
    for s in st:
        print("   ", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items()}, flush=True)

# ICP
for n in (4000, 100000):
    src, tgt, T_gt = synth.make_pair(n, seed=5)
    nrm = icp.normals_from_cov(tgt["cov6"])
    nrm_o = O.normals_from_cov(np.stack([tgt["cov6"][:, [0,1,2]], tgt["cov6"][:, [1,3,4]], tgt["cov6"][:, [2,4,5]]], 1).astype(np.float64))
    print("normals max |abs dot| deficit", np.max(1 - np.abs(np.sum(nrm * nrm_o, 1))))
    for kind in (0, 1):
        t = time.time(); w = O.icp(src["xyz"], tgt["xyz"], nrm_o, np.eye(4), kind=kind, max_corr=0.3, max_iter=30); to = time.time() - t
        t = time.time(); r = icp.registration_icp_arrays(src["xyz"], tgt["xyz"], nrm_o, np.eye(4), kind=kind, max_corr=0.3, max_iter=30); tg = time.time() - t
        print(f"icp n={n} kind={kind}: |T-T_oracle|_F={np.linalg.norm(r['transformation']-w['transformation']):.3e} |T-T_gt|_F={np.linalg.norm(r['transformation']-T_gt):.3e} "
              f"fit {r['fitness']:.5f}/{w['fitness']:.5f} rmse {r['inlier_rmse']:.6f}/{w['inlier_rmse']:.6f} it {r['iterations']}/{w['iterations']} "
              f"oracle {to:.2f}s gpu {tg:.2f}s kernels {r['ms_iters']:.2f}ms/{r['iter_kernels']} build {r['ms_build']:.2f}ms", flush=True)
print("DONE")
