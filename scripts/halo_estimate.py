"""Estimate of the ghost set of one rank of the spatially partitioned HEM level at 40 M splats on 8 ranks: the cloud of
synth.make_cloud (uniform in a cube, radius R = 3 max(s), s = exp(N(-2.5, 0.5))), slabs along x, cells of 16 components.  A rank's
ghosts are the components of OTHER slabs in the grid cells its parents' search boxes touch (k_mark_cells).  Monte Carlo over the
parents of one interior rank; 3-D difference array over the neighbouring slabs' cells."""
import numpy as np
import sys
n, world, rho, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000), 8, 3.0, 45
dens = 50000 / 27.0                      # synth.half_extent: 50 k splats in a cube of side 3
side = (n / dens) ** (1 / 3)
slab = side / world
c = (16 / dens) ** (1 / 3)
rng = np.random.default_rng(0)
P = int(n / world / rho)                 # parents of the rank
R = 3.0 * np.exp(rng.normal(-2.5, 0.5, (P, 3))).max(1)
x = rng.uniform(0, slab, P); y = rng.uniform(0, side, P); z = rng.uniform(0, side, P)
gy = int(side / c) + 1
tot_cells = 0
for sgn in (0, 1):                       # neighbour slab on the low / high side
    a = x if sgn == 0 else slab - x      # distance to that boundary
    m = R > a
    depth = np.minimum(R[m] - a[m], slab * (world - 1))
    d1 = np.minimum((depth / c).astype(int) + 1, int(slab * (world - 1) / c))
    y0 = np.clip(((y[m] - R[m]) / c).astype(int), 0, gy - 1); y1 = np.clip(((y[m] + R[m]) / c).astype(int), 0, gy - 1) + 1
    z0 = np.clip(((z[m] - R[m]) / c).astype(int), 0, gy - 1); z1 = np.clip(((z[m] + R[m]) / c).astype(int), 0, gy - 1) + 1
    gd = int(d1.max()) + 1
    D = np.zeros((gd + 1, gy + 1, gy + 1), np.int32)
    for sx, xs in ((1, np.zeros_like(d1)), (-1, d1)):
        for sy, ys in ((1, y0), (-1, y1)):
            for sz, zs in ((1, z0), (-1, z1)):
                np.add.at(D, (xs, ys, zs), sx * sy * sz)
    cov = D.cumsum(0).cumsum(1).cumsum(2)[:gd, :gy, :gy] > 0
    per_layer = cov.reshape(gd, -1).mean(1)
    tot_cells += cov.sum()
    print(f"side {sgn}: parents reaching over {m.sum()} of {P}; covered fraction of the cross-section by depth (units):",
          " ".join(f"{(k + 1) * c:.2f}:{v:.2f}" for k, v in enumerate(per_layer) if k % 2 == 1 and v > 0.001))
ghosts = tot_cells * 16
row = 64 + 4 * F + 8
print(f"cloud side {side:.2f}, slab {slab:.2f}, cell {c:.3f}; ghosts of an interior rank ~ {ghosts / 1e6:.2f} M = {ghosts / (n / world):.0%} of its own {n / world / 1e6:.1f} M; "
      f"halo rows received {ghosts * row / 1e6:.0f} MB at {row} B per row")

# the same cloud cut into 2 x 2 x 2 blocks (parallel.block_of): a block has three interior faces; coverage of the cells outside the block
bs = side / 2
gb = int(bs / c) + 1
pad = int(3.6 / c) + 1
x = rng.uniform(0, bs, P); y = rng.uniform(0, bs, P); z = rng.uniform(0, bs, P)       # block [0, bs]^3, neighbours on the + side of each axis
lo = [np.clip(((v - R) / c).astype(int), 0, gb + pad - 1) for v in (x, y, z)]
hi = [np.clip(((v + R) / c).astype(int), 0, gb + pad - 1) + 1 for v in (x, y, z)]
m = (x + R > bs) | (y + R > bs) | (z + R > bs)
G = gb + pad + 1
D = np.zeros((G + 1, G + 1, G + 1), np.int32)
for sx, xs in ((1, lo[0][m]), (-1, hi[0][m])):
    for sy, ys in ((1, lo[1][m]), (-1, hi[1][m])):
        for sz, zs in ((1, lo[2][m]), (-1, hi[2][m])):
            np.add.at(D, (xs, ys, zs), sx * sy * sz)
cov = D.cumsum(0).cumsum(1).cumsum(2)[:G, :G, :G] > 0
inside = np.zeros_like(cov); inside[:gb, :gb, :gb] = True
gh = (cov & ~inside).sum() * 16
print(f"2 x 2 x 2 blocks of side {bs:.2f}: ghosts ~ {gh / 1e6:.2f} M = {gh / (n / world):.0%} of the rank's own; halo rows received {gh * row / 1e6:.0f} MB")
