"""Wall time of one asynchronous level at 50 k / 200 k / 556 k splats (host clock, level 0 resident, context warm): a short ladder for
A/B runs of the small levels.  usage: small_levels.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussiansplattingregistration_amd import hem, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
out = []
for n in (50_000, 200_000, 556_000):
    c = synth.make_cloud_torch(n, seed=100)
    with hem.HemMixture() as m:
        ts = []
        for _ in range(reps):
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)
            torch.cuda.synchronize(); t = time.perf_counter()
            m.run_level()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        out.append(f"{n // 1000} k {float(np.median(ts[2:])) * 1e3:.3f}")
print("  ".join(out))
