"""Host-side cost of the calls around one cloud's three HEM levels (the bench's hem_levels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingregistration_amd import hem, synth
from gaussiansplattingregistration_amd.models.point_cloud import PointCloud

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
c = synth.make_cloud_torch(n, seed=100)
m = hem.HemMixture()
sync = torch.cuda.synchronize
for rep in range(3):
    m.set_rng("glibc", 1, 0)
    t = {}
    def tm(k, f):
        sync(); t0 = time.perf_counter(); r = f(); sync(); t[k] = t.get(k, 0.0) + (time.perf_counter() - t0) * 1e3; return r
    T0 = time.perf_counter()
    tm("set_level0", lambda: m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True))
    lvl_ms = 0.0
    for l in range(3):
        tm("run_level", lambda: m.run_level())
        st = tm("stats", lambda: m.stats()); lvl_ms += st["ms_level"]
        d = tm("get_level", lambda: m.get_level(as_torch=True))
        tm("PointCloud", lambda: PointCloud(xyz32=d["xyz"], cov6=d["cov6"]))
    sync(); tot = (time.perf_counter() - T0) * 1e3
    print(f"rep{rep} total {tot:.3f} ms, sum ms_level {lvl_ms:.3f};", {k: round(v, 3) for k, v in t.items()}, flush=True)
