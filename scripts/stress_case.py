import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["STRESS_PATHOLOGIES"] = "1"
import numpy as np
import stress_parity as sp
rng = np.random.default_rng(607)
for k in range(179):
    desc, c, p = sp.make_case(rng, True)
print(desc)
from gaussiansplattingregistration_amd import hem
from oracle import oracle as O
want, wst = O.hem(c, 1, **p)
with hem.HemMixture(hem_reduction=p["rho"], distance_delta=p["delta"], color_delta=p["kappa"], decay_rate=p["tau"]) as m:
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"]); m.run_level(); st, got = m.stats(), m.get_level()
print(st["dropped"], wst[0]["dropped"], sp.unmatched_rows_near_singular(got, want[0]))
ng, nw, ok = sp.unmatched_rows_near_singular(got, want[0], lim=1e-5); print("lim 1e-5:", ng, nw, ok)
for lim in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    print("lim", lim, sp.unmatched_rows_near_singular(got, want[0], lim=lim))
