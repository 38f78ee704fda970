import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from gaussiansplattingregistration_amd import hem, synth
n = 5_000_000
m = hem.HemMixture()
m.set_timing(2)
for seed, shape in ((100, "aniso"), (300, "aniso"), (100, "iso"), (400, "clustered")):
    c = synth.make_cloud_torch(n, seed=seed, shape=shape)
    for borrow in (True,):
        for rep in range(3):
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=borrow)
            m.run_level()
            st = m.stats()
        print(f"seed {seed} {shape} max pairs of a parent {st['max_pairs_of_a_parent']} heavy parents {st['heavy_parents']}: level {st['ms_level']:.3f} select {st['ms_k_select']:.3f} mstep {st['ms_k_mstep']:.3f} phases grid {st['ms_grid']:.3f} sel {st['ms_select']:.3f} sum {st['ms_sumlw']:.3f} mstep {st['ms_mstep']:.3f} flags {st['ms_flags']:.3f} pairs {st['pairs']} orphans {st['orphans']}", flush=True)
