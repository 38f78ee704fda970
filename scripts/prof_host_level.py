"""Host-side cost of the calls bench.py makes per level (new_output / run_level / stats / get_level), 5 M cloud, 3 levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingregistration_amd import hem, synth
c = synth.make_cloud_torch(5_000_000, seed=100)
m = hem.HemMixture()
sync = torch.cuda.synchronize
for rep in range(4):
    m.set_rng("glibc", 1, 0)
    sync(); t0 = time.perf_counter()
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)
    sync(); t_set = time.perf_counter() - t0
    rows = []
    for l in range(3):
        sync(); a = time.perf_counter()
        o = m.new_output(); b = time.perf_counter()
        m.run_level(out=o); sync(); cc = time.perf_counter()
        st = m.stats(); d = time.perf_counter()
        lv = m.get_level(as_torch=True); e = time.perf_counter()
        rows.append((b - a, cc - b, st["ms_level"], d - cc, e - d))
    print(f"rep{rep} set_level0 {t_set*1e3:.3f} ms | " + " | ".join(f"L{i+1}: new_output {r[0]*1e3:.3f} run_level {r[1]*1e3:.3f} (device {r[2]:.3f}) stats {r[3]*1e3:.3f} get_level {r[4]*1e3:.3f}" for i, r in enumerate(rows)), flush=True)
