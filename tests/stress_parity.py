"""Randomised parity sweep (test infrastructure: the oracle is the checker).  Many small clouds of random size, shape, density,
SH degree and HEM parameters; level 1 of every cloud on the GPU against oracle/hem_oracle.cpp: the discrete outcomes (parents,
accepted pairs, orphans, dropped, level size) must be EQUAL and the components within 1e-4.  One documented exception is counted
apart ("edge"): the validity erase (det <= 0 of a MERGED covariance, mixture.cpp:262-274) of a near-singular merged needle can go
either way with the summation order of the M-step -- parents, pairs and orphans equal, `dropped` off by one (seen in 6 of 400
clouds, all "needles" with delta = 2).  Not collected by pytest (no test_ prefix): run it through gpurun when the selection path
changes --  python tests/stress_parity.py [cases] [seed]   (STRESS_PATHOLOGIES=1: a handful of NaN / inf / degenerate components in
every cloud)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gaussiansplattingregistration_amd import hem, synth
from oracle import oracle as O

TOL = 1e-4


def rel(a, b):
    """max |a - b| over max |b| on the finite entries; inf when NaN / inf sit in different places"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb) or not np.array_equal(np.isnan(a), np.isnan(b)):
        return float("inf")
    if not fa.any():
        return 0.0
    return float(np.max(np.abs(a[fa] - b[fb])) / (np.max(np.abs(b[fb])) + 1e-30))


def unmatched_rows_near_singular(got, want, lim=1e-3):
    """The validity erase (det <= 0 of a MERGED covariance in float32, mixture.cpp:262-274) dropped different rows on the two sides.  Match the rows of the
    two levels on their positions (nearest neighbour within 1e-4 of the scene's extent); -> (rows only in `got`, rows only in `want`, all of them
    NEAR-SINGULAR): |det| in float64 of the float32 entries below `lim` x the product of the diagonal -- a determinant that is cancellation noise in
    float32, whose sign follows the summation order of the M-step -- or not finite.  Rows with non-finite positions must be equally many."""
    from scipy.spatial import cKDTree
    gx, wx = np.asarray(got["xyz"], np.float64), np.asarray(want["xyz"], np.float64)
    fg, fw = np.isfinite(gx).all(1), np.isfinite(wx).all(1)
    if int((~fg).sum()) != int((~fw).sum()):
        return int((~fg).sum()), int((~fw).sum()), False
    ig, iw = np.where(fg)[0], np.where(fw)[0]
    if len(ig) == 0 or len(iw) == 0:
        return len(ig), len(iw), False
    scale = np.abs(wx[iw]).max() + 1e-30
    dg, _ = cKDTree(wx[iw]).query(gx[ig], k=1)
    dw, _ = cKDTree(gx[ig]).query(wx[iw], k=1)
    only_g, only_w = ig[dg > 1e-4 * scale], iw[dw > 1e-4 * scale]

    def near_singular(c6):
        c = np.asarray(c6, np.float64).reshape(-1, 6)
        det = (-c[:, 2] * c[:, 2] * c[:, 3] + 2 * c[:, 1] * c[:, 2] * c[:, 4] - c[:, 0] * c[:, 4] * c[:, 4] - c[:, 1] * c[:, 1] * c[:, 5] + c[:, 0] * c[:, 3] * c[:, 5])
        ref = np.abs(c[:, 0] * c[:, 3] * c[:, 5])
        return ~np.isfinite(det) | ~np.isfinite(ref) | (np.abs(det) <= lim * ref)
    ok = bool(near_singular(np.asarray(got["cov6"])[only_g]).all()) and bool(near_singular(np.asarray(want["cov6"])[only_w]).all())
    return len(only_g), len(only_w), ok


def make_case(rng, pathologies=None):
    if pathologies is None:
        pathologies = bool(os.environ.get("STRESS_PATHOLOGIES"))
    n = int(rng.integers(2000, 90000))
    shape = rng.choice(["iso", "aniso", "needles", "clustered"])
    deg = int(rng.choice([0, 1, 2, 3]))
    h = synth.half_extent(n) * float(rng.choice([0.5, 1.0, 1.0, 2.0]))          # denser / the bench density / sparser
    seed = int(rng.integers(1 << 30))
    c = synth.make_cloud(n, seed=seed, h=h, sh_degree=deg, shape="aniso" if shape == "aniso" else "iso")
    if shape == "needles":                                                       # a third of the splats squashed: kappa 1e3 .. 1e6
        k = n // 3
        idx = rng.choice(n, k, replace=False)
        s = np.exp(rng.normal(-2.5, 0.5, (k, 3)))
        s[:, 0] *= rng.choice([1e-2, 3e-2, 1e-3], k)
        s[: k // 2, 1] *= 0.05
        q = rng.normal(size=(k, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        L = synth._quat_to_rot(q) * s[:, None, :]
        C = (L @ L.transpose(0, 2, 1)).astype(np.float32)
        c["cov6"][idx] = C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
    if shape == "clustered":                                                     # half of the cloud pulled into a few tight clumps, a few far outliers
        k = n // 2
        centres = rng.uniform(-h, h, (5, 3))
        c["xyz"][:k] = (centres[rng.integers(0, 5, k)] + rng.normal(0, 0.05 * h, (k, 3))).astype(np.float32)
        c["xyz"][-3:] *= 50.0
    if pathologies:              # a handful of broken components in a cloud of ordinary size
        def pick(m):
            return rng.choice(n, m, replace=False)
        if rng.random() < 0.5: c["cov6"][pick(20)] = 0.0
        if rng.random() < 0.5: c["cov6"][pick(20)] = np.array([1, 0, 0, 1, 0, -1], np.float32) * 0.01
        if rng.random() < 0.4: c["cov6"][pick(10), int(rng.integers(6))] = np.nan
        if rng.random() < 0.4: c["xyz"][pick(10), int(rng.integers(3))] = np.nan
        if rng.random() < 0.3: c["xyz"][pick(5), int(rng.integers(3))] = np.inf
        if rng.random() < 0.4: c["cov6"][pick(30)] *= np.float32(1e8)
        if rng.random() < 0.4: c["cov6"][pick(30)] *= np.float32(1e-12)
        if rng.random() < 0.3: c["opacity"][pick(20)] = np.nan
        if rng.random() < 0.3: i = pick(200); c["xyz"][i] = c["xyz"][rng.choice(n, 200)]
    params = dict(rho=float(rng.choice([2.0, 3.0, 3.0, 5.0])), delta=float(rng.choice([2.0, 3.0, 3.0, 4.0])),
                  kappa=float(rng.choice([1.5, 2.5, 2.5, 4.0])), tau=float(rng.choice([0.5, 1.0, 1.0, 2.0])))
    return dict(n=n, shape=shape, deg=deg, h=round(h, 3), seed=seed, **params), c, params


def sweep(cases=40, seed=2026, log=print, pathologies=None):
    """-> (equal, edge, bad): `edge` = exactly the documented class (parents, pairs and orphans EQUAL, `dropped` and the level
    size off by <= 2 -- 8 with STRESS_PATHOLOGIES), anything else is `bad`.  tests/test_stress_gpu.py runs a fixed-seed slice."""
    if pathologies is None:
        pathologies = bool(os.environ.get("STRESS_PATHOLOGIES"))
    rng = np.random.default_rng(seed)
    bad = edge = 0
    t0 = time.time()
    for k in range(cases):
        desc, c, p = make_case(rng, pathologies)
        want, wst = O.hem(c, 1, **p)
        with hem.HemMixture(hem_reduction=p["rho"], distance_delta=p["delta"], color_delta=p["kappa"], decay_rate=p["tau"]) as m:
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            m.run_level()
            st, got = m.stats(), m.get_level()
        a = (st["parents"], st["pairs"], st["orphans"], st["dropped"], got["xyz"].shape[0])
        b = (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"], want[0]["xyz"].shape[0])
        err = max(rel(got[f], want[0][f]) for f in ("xyz", "color", "cov6", "opacity", "sh")) if a[4] == b[4] else float("inf")
        ok = a == b and err < TOL
        lim = 8 if pathologies else 2     # (merges with a 1e8-scaled needle: the determinant is cancellation noise)
        edge_case = (not ok) and a[:3] == b[:3] and abs(a[3] - b[3]) <= lim and abs(a[4] - b[4]) <= lim
        if (not ok) and (not edge_case) and a[:3] == b[:3]:
            # more rows than that (round 6, seed 607 case 178: 3 582 needles, rho 2, delta 4 and thirty 1e8-scaled covariances -- every parent merges ~240
            # children and a sixth of the merges swallow a giant: 299 against 341 rows dropped): the same class iff EVERY row that one side kept and the
            # other dropped is near-singular (its float32 determinant is cancellation noise)
            ng, nw, singular = unmatched_rows_near_singular(got, want[0])
            edge_case = singular and (ng + nw) > 0
            desc = dict(desc, unmatched=(ng, nw), all_near_singular=singular)
        edge += 1 if edge_case else 0
        bad += 0 if (ok or edge_case) else 1
        log(f"{'ok  ' if ok else ('edge' if edge_case else 'FAIL')} {k:3d} {desc}  gpu {a}  oracle {b}  irregular {st['irregular']}  max rel {err:.2e}")
    log(f"{cases - bad - edge} of {cases} cases equal the oracle, {edge} differ by a borderline validity erase, {bad} FAIL ({time.time() - t0:.0f} s)")
    return cases - bad - edge, edge, bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    _, _, bad = sweep(cases, seed, log=lambda s: print(s, flush=True))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
