import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gaussiansplattingregistration_amd import hem, synth
dev = torch.device("cuda", 0)
m = hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS)
for shape, seed in (("iso", 100), ("aniso", 300), ("clustered", 400), ("iso", 100), ("clustered", 400)):
    c = synth.make_cloud_torch(5_000_000, seed=seed, device=dev, shape=shape)
    for rep in range(5):
        m.set_timing(2 if rep == 4 else 1)
        m.set_rng("glibc", 1, 0)
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)
        m.run_level()
        st = m.stats()
        print(shape, rep, "ms_level %.3f sched %d rt %d ksel %.3f kmst %.3f" % (st["ms_level"], st["schedule"], st["round_trips"], st["ms_k_select"], st["ms_k_mstep"]),
              {k: round(st[k], 3) for k in ("ms_grid", "ms_select", "ms_sumlw", "ms_mstep", "ms_flags")} if rep == 4 else "", flush=True)
    del c
