#!/usr/bin/env python3
"""How much would a workgroup-shared LDS tile of candidate records buy k_select?

For the bench cloud (SURVEY 8(d) generator, constant density) at level 1: parents are grouped into tiles of T consecutive
parents along the Z-order curve the kernel already processes them in; a tile's LDS image would hold the UNION of the
candidate sets (all components within the query radius R_s = delta sqrt(lambda_max) of some parent of the tile, what
stage 1 streams).  Reported per tile size: the union (records and bytes at 16 B), the sum of the per-parent sets, and
their ratio = how often a staged record would be re-used; then the same for the ACCEPTED children (the M-step's gathers).  CPU only (scipy cKDTree); run on a sub-box of the cloud.

    python scripts/tile_reuse.py [n_splats=400000]  > profiles/archive/r02_tile_reuse.txt
"""
import os
import sys

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingregistration_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
c = synth.make_cloud(n, seed=100)
xyz = c["xyz"].astype(np.float64)
lam = O.eigenvalues(c["cov6"])[:, 2].astype(np.float64)
R = 3.0 * np.sqrt(lam)
par = O.parent_flags(n, 3.0).astype(bool)
tree = cKDTree(xyz)
pidx = np.nonzero(par)[0]
# Z-order of the parents over a 1024^3 lattice
h = c["h"]
q = np.clip(((xyz[pidx] + h) / (2 * h) * 1023).astype(np.int64), 0, 1023)


def spread(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
pz = pidx[np.argsort(key, kind="stable")]
rng = np.random.default_rng(0)
def accepted_sets(ps):
    """Children parent s accepts (float64 restatement of the gates of mixture.cpp:102-137: radius, colour, KL, parent rule)."""
    out = []
    C = np.stack([c["cov6"][:, [0, 1, 2]], c["cov6"][:, [1, 3, 4]], c["cov6"][:, [2, 4, 5]]], 1).astype(np.float64)
    col = c["color"].astype(np.float64)
    for p in ps:
        cand = np.array(tree.query_ball_point(xyz[p], R[p]))
        cand = cand[np.linalg.norm(xyz[cand] - xyz[p], axis=1) < R[p]]
        Pinv = np.linalg.inv(C[p])
        d = xyz[cand] - xyz[p]
        smd = np.einsum("ni,ij,nj->n", d, Pinv, d)
        tr = np.einsum("ij,nji->n", Pinv, C[cand])
        kld = 0.5 * (smd + tr - 3 - np.log(np.linalg.det(C[cand]) / np.linalg.det(C[p])))
        ok = (np.linalg.norm(col[cand] - col[p], axis=1) <= 2.5 ** 2 / 2) & (kld <= 4.5) & (~par[cand] | (cand == p))
        out.append(set(cand[ok].tolist()))
    return out


print(f"# {n} splats, {len(pidx)} parents, density {n / (2 * h) ** 3:.0f} / unit^3, mean R {R[par].mean():.3f}, "
      f"median {np.median(R[par]):.3f}, 99th pct {np.percentile(R[par], 99):.3f}")
print("# tile = T consecutive parents in Z-order; union / sum of the parents' candidate sets (sphere of radius R_s)")
print(f"{'T':>5} {'sum of sets':>12} {'union':>9} {'re-use':>7} {'LDS bytes (16 B A-records)':>27} {'fits 160 KiB':>13} {'tiles whose union fits':>23}")
for T in (4, 8, 16, 32, 64, 128, 256):
    starts = rng.choice(max(1, len(pz) - T), size=min(300, max(1, len(pz) // T)), replace=False)
    sums, unions = [], []
    for s in starts:
        ps = pz[s:s + T]
        sets = tree.query_ball_point(xyz[ps], R[ps])
        sums.append(sum(len(x) for x in sets))
        unions.append(len(set().union(*map(set, sets))))
    sums, unions = np.array(sums), np.array(unions)
    fits = (unions * 16 <= 160 * 1024).mean()
    print(f"{T:5d} {sums.mean():12.0f} {unions.mean():9.0f} {sums.mean() / unions.mean():7.2f} {unions.mean() * 16:27.0f} "
          f"{'yes' if unions.mean() * 16 <= 160 * 1024 else 'no':>13} {100 * fits:22.0f}%")

print()
print("# the same for the ACCEPTED children (what the M-step gathers: one 64-B geometry record + one 192-B SH row per pair)")
print(f"{'T':>5} {'pairs':>9} {'distinct':>9} {'re-use':>7} {'LDS bytes (256 B per child)':>28}")
for T in (8, 16, 32, 64):
    starts = rng.choice(max(1, len(pz) - T), size=40, replace=False)
    sums, unions = [], []
    for s0 in starts:
        sets = accepted_sets(pz[s0:s0 + T])
        sums.append(sum(len(x) for x in sets))
        unions.append(len(set().union(*sets)))
    sums, unions = np.array(sums), np.array(unions)
    print(f"{T:5d} {sums.mean():9.0f} {unions.mean():9.0f} {sums.mean() / unions.mean():7.2f} {unions.mean() * 256:28.0f}")
