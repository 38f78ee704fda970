"""BASELINE.json configs C2 (2 x 1 M splats, 3 HEM levels + point-to-plane ICP) and C3 (2 x 5 M splats, SH degree 3,
4-level coarse-to-fine registration) end to end on one MI355X, through the same front end bench.py drives.

C2 is small enough for the CPU oracle to run beside it: level 1 must equal the oracle's discrete outcome exactly
(parents, accepted pairs, orphans) and its components to 1e-4; levels 2-3 are compared cascade-free (every level
recomputed from the ORACLE's previous level) and end to end through the global moments of the mixture; the ICP chain on
the GPU's own level lists is compared with the oracle's chain on the same arrays (final transform <= 1e-5 Frobenius).
C3 (the configuration the metric is quoted on) is compared with the reference's VALUES at its own size: level 1 of the bench's 5 M-splat
cloud against tests/golden/hem_5m_digest.npz (counts, every row's parent flag, float64 global moments, per-block column sums over ALL
rows, 10 000 sampled rows with every array on the row's own scale; made by tests/golden/make_golden_5m.py in the build container), levels
2-3 end to end through counts and global moments; then size-independent properties per level on the bench's own PAIR, the distance to the
ground-truth motion, and the oracle's ICP on all four entries of the schedule, the 5 M x 5 M one included.

Tolerances (BASELINE.json north_star): mixture moments 1e-4 relative, transforms 1e-5 Frobenius.  "Relative" for a
mean of zero-mean fields (colour, SH) is taken against the RMS magnitude of the field, for the weighted mean position
against the cloud's extent, for the covariance against its largest entry.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

HEM_PARAMS = dict(hem_reduction=3.0, distance_delta=3.0, color_delta=2.5, decay_rate=1.0)
ITER_VALUES = [50, 30, 20, 10]
MAX_CORR = [0.5, 0.3, 0.2, 0.1]


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def global_moments(lv):
    """Global moments of a mixture level: sum of weights, weighted mean, total (within + between) weighted covariance,
    weighted mean colour / opacity / SH, and the RMS magnitudes the relative errors refer to."""
    w = _np(lv["weight"]).astype(np.float64)
    W = w.sum()
    x = _np(lv["xyz"]).astype(np.float64)
    c6 = _np(lv["cov6"]).astype(np.float64)
    mean = (w[:, None] * x).sum(0) / W
    d = x - mean
    outer = np.stack([d[:, 0] * d[:, 0], d[:, 0] * d[:, 1], d[:, 0] * d[:, 2], d[:, 1] * d[:, 1], d[:, 1] * d[:, 2], d[:, 2] * d[:, 2]], 1)
    cov = (w[:, None] * (c6 + outer)).sum(0) / W
    col = _np(lv["color"]).astype(np.float64)
    op = _np(lv["opacity"]).astype(np.float64)
    sh = _np(lv["sh"]).astype(np.float64)
    return {"W": W, "mean": mean, "cov": cov,
            "color": (w[:, None] * col).sum(0) / W, "opacity": (w * op).sum() / W, "sh": (w[:, None] * sh).sum(0) / W,
            "rms_color": np.sqrt((w[:, None] * col * col).sum() / W / 3), "rms_opacity": np.sqrt((w * op * op).sum() / W),
            "rms_sh": np.sqrt((w[:, None] * sh * sh).sum() / W / max(1, sh.shape[1])), "extent": np.abs(x).max()}


def assert_moments_close(got, want, tag, tol=1e-4):
    g, o = global_moments(got), global_moments(want)
    assert abs(g["W"] - o["W"]) <= tol * o["W"], (tag, "sum of weights", g["W"], o["W"])
    assert np.abs(g["mean"] - o["mean"]).max() <= tol * o["extent"], (tag, "weighted mean", g["mean"], o["mean"])
    assert np.abs(g["cov"] - o["cov"]).max() <= tol * np.abs(o["cov"]).max(), (tag, "weighted covariance", g["cov"], o["cov"])
    assert np.abs(g["color"] - o["color"]).max() <= tol * o["rms_color"], (tag, "mean colour", g["color"], o["color"])
    assert abs(g["opacity"] - o["opacity"]) <= tol * o["rms_opacity"], (tag, "mean opacity", g["opacity"], o["opacity"])
    assert np.abs(g["sh"] - o["sh"]).max() <= tol * o["rms_sh"], (tag, "mean SH", float(np.abs(g["sh"] - o["sh"]).max()), o["rms_sh"])


def _rel(a, b):
    a, b = _np(a).astype(np.float64), _np(b).astype(np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)) if a.size else 0.0


def assert_rows_close(got, want, tag, tol=1e-4):
    """Per-component check on each component's OWN scale (VERDICT r03 "weak" 5: max |d| over max |field| lets a small component be
    off by far more than 1e-4 of itself; VERDICT r05 "weak" 3: every array, not positions and covariance diagonals only).  ALL SIX
    covariance entries against the component's own trace, positions against its own extent sqrt(trace) (absolute floor: 1e-3 of the
    field's median trace -- a degenerate component is not asked for more than float32 can give the sum it came from); colour, SH,
    opacity and weight against the row's own largest |entry| of that array, floored at 5 % of the array's RMS: these are zero-mean fields,
    and an entry that cancels far below the array's scale is a float32 sum of terms tens to thousands of times its size -- 1e-4 of the
    entry itself would ask for more digits than float32 carried through the sum (seen at 1 M rows: a merged opacity of -3.98e-4 from raw
    opacities of magnitude 2, GPU and oracle 1.6e-7 apart: two summation orders of the same float32 terms)."""
    gx, wx = _np(got["xyz"]).astype(np.float64), _np(want["xyz"]).astype(np.float64)
    gc, wc = _np(got["cov6"]).astype(np.float64), _np(want["cov6"]).astype(np.float64)
    assert gx.shape == wx.shape and gc.shape == wc.shape, tag
    tr = wc[:, 0] + wc[:, 3] + wc[:, 5]
    scale = np.maximum(tr, 1e-3 * np.median(tr))
    e_cov = np.abs(gc - wc).max(1) / scale
    e_pos = np.abs(gx - wx).max(1) / np.sqrt(scale)
    i, j = int(np.argmax(e_cov)), int(np.argmax(e_pos))
    assert e_cov[i] <= tol, (tag, "covariance (all six entries), per component", i, float(e_cov[i]), gc[i], wc[i])
    assert e_pos[j] <= tol, (tag, "position, per component", j, float(e_pos[j]), gx[j], wx[j])
    for f in ("color", "sh", "opacity", "weight"):
        if f not in got or f not in want:
            continue
        g_, w_ = _np(got[f]).astype(np.float64).reshape(len(wx), -1), _np(want[f]).astype(np.float64).reshape(len(wx), -1)
        if w_.shape[1] == 0:
            continue
        e = np.abs(g_ - w_).max(1) / np.maximum(np.abs(w_).max(1), 5e-2 * np.sqrt((w_ * w_).mean()))
        k = int(np.argmax(e))
        assert e[k] <= tol, (tag, f + ", per component", k, float(e[k]), g_[k][:6], w_[k][:6])


def level_properties(prev, cur, st, dropped, h, tag):
    """Size-independent properties of one level: bookkeeping, weight conservation, mean preservation by moment matching,
    finite values, positive-definite covariances."""
    n_in = prev["xyz"].shape[0]
    assert st["parents"] == int(_np(prev["is_parent"]).sum()), tag
    assert cur["xyz"].shape[0] == st["parents"] + st["orphans"] - dropped, tag
    assert abs(st["parents"] - n_in / 3) < 0.02 * n_in, (tag, st["parents"], n_in)          # parent probability 1 / rho
    wp, wc = prev["weight"].double(), cur["weight"].double()
    assert abs(float(wc.sum()) - float(wp.sum())) <= 1e-4 * float(wp.sum()), (tag, float(wc.sum()), float(wp.sum()))
    m0 = (wp[:, None] * prev["xyz"].double()).sum(0) / wp.sum()
    m1 = (wc[:, None] * cur["xyz"].double()).sum(0) / wc.sum()
    assert float((m0 - m1).abs().max()) <= 1e-4 * h, (tag, m0, m1)
    for f in ("xyz", "color", "cov6", "opacity", "sh", "weight"):
        assert bool(torch.isfinite(cur[f]).all()), (tag, f)
    c = cur["cov6"].double()
    det = (-c[:, 2] * c[:, 2] * c[:, 3] + 2 * c[:, 1] * c[:, 2] * c[:, 4] - c[:, 0] * c[:, 4] * c[:, 4]
           - c[:, 1] * c[:, 1] * c[:, 5] + c[:, 0] * c[:, 3] * c[:, 5])
    assert bool((det > 0).all()), tag


def gpu_level_lists(src, tgt, levels=3):
    """What bench.py's step does for the HEM half: cloud 1 then cloud 2 on ONE libc rand() stream (a fresh reference
    process, qt_gaussian_mixture.py:55,79).  Returns per cloud the list [level 0, level 1, ...] with state, and the stats."""
    from gaussiansplattingregistration_amd import hem
    out, stats = [], []
    with hem.HemMixture(rng_mode="glibc", **HEM_PARAMS) as m:
        m.set_rng("glibc", 1, 0)
        for c in (src, tgt):
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            lv, st = [m.get_level(as_torch=True, with_state=True)], []
            for _ in range(levels):
                _, dropped = m.run_level()
                s = m.stats()
                s["dropped_now"] = dropped
                st.append(s)
                lv.append(m.get_level(as_torch=True, with_state=True))
            out.append(lv)
            stats.append(st)
    return out, stats


def gpu_multiscale(lists_src, lists_tgt, kind_plane=True):
    """Coarse-to-fine ICP over the level lists (qt_multiscale_registrator.py:197-236).  Returns the per-level results."""
    from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
    from gaussiansplattingregistration_amd.utils import local_registration_util as lru
    est = lru.get_estimation(lru.LocalRegistrationType.ICP_Point_To_Plane if kind_plane else lru.LocalRegistrationType.ICP_Point_To_Point,
                             lru.RobustLoss(0))
    T = np.eye(4)
    res, clouds = [], []
    for k in range(len(ITER_VALUES)):
        s_l, t_l = lists_src[-(k + 1)], lists_tgt[-(k + 1)]
        s = PointCloud(xyz32=s_l["xyz"], cov6=s_l["cov6"])
        t = PointCloud(xyz32=t_l["xyz"], cov6=t_l["cov6"])
        t.estimate_normals()
        r = lru.registration_icp(s, t, MAX_CORR[k], T, est, lru.get_convergence_criteria(1e-6, 1e-6, ITER_VALUES[k]))
        res.append((T.copy(), r))
        clouds.append((s, t))
        T = r.transformation
    return res, clouds


def make_pair_torch(n, seed, angle_deg, shift):
    """target = synthetic cloud (SURVEY 8d generator, on the device), source = inv(T_gt) * target + 0.002 jitter."""
    from gaussiansplattingregistration_amd import synth
    dev = torch.device("cuda", 0)
    tgt = synth.make_cloud_torch(n, seed=seed, device=dev)
    T_gt = synth.rigid_transform(angle_deg, (1, 1, 1), shift * tgt["h"] * np.array([1.0, -1.0, 0.5]))
    src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
    gen = torch.Generator(device=dev).manual_seed(7)
    src["xyz"] = src["xyz"] + torch.randn(src["xyz"].shape, device=dev, generator=gen) * 0.002
    src = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in src.items()}
    return src, tgt, T_gt


def test_c2_2x1m_three_levels_and_point_to_plane_icp(oracle):
    """BASELINE configs[1]: 2 x 1 M splats, 3 HEM mixture levels + point-to-plane ICP, one MI355X."""
    from gaussiansplattingregistration_amd import hem
    n = 1_000_000
    src, tgt, T_gt = make_pair_torch(n, seed=100, angle_deg=2.0, shift=0.01)
    lists, stats = gpu_level_lists(src, tgt)
    h = tgt["h"]
    for ci in range(2):
        for k in range(3):
            level_properties(lists[ci][k], lists[ci][k + 1], stats[ci][k], stats[ci][k]["dropped_now"], h, ("C2", ci, k))
    sizes = [lv["xyz"].shape[0] for lv in lists[0]]
    assert sizes[0] == n and all(0.30 * a < b < 0.37 * a for a, b in zip(sizes, sizes[1:])), sizes

    # ---- HEM against the oracle on the first cloud (the one that starts the rand() stream)
    host = {k: _np(src[k]) for k in ("xyz", "color", "cov6", "opacity", "sh")}
    o = oracle.HemOracle(host["xyz"], host["color"], host["cov6"], host["opacity"], host["sh"])
    olv, ost = [o.level(0)], []
    for k in range(3):
        o.run_level()
        ost.append(o.stats())
        olv.append(o.level(k + 1))
    o.close()
    # level 0 state: the same parent flags
    assert np.array_equal(_np(lists[0][0]["is_parent"]), olv[0]["is_parent"])
    # level 1: the discrete outcome is exact, the components agree to 1e-4
    g1, s1 = lists[0][1], stats[0][0]
    assert (s1["parents"], s1["pairs"], s1["orphans"], s1["dropped_now"]) == (ost[0]["parents"], ost[0]["pairs"], ost[0]["orphans"], ost[0]["dropped"])
    assert g1["xyz"].shape[0] == olv[1]["xyz"].shape[0]
    for f in ("xyz", "color", "cov6", "opacity", "sh", "weight"):
        assert _rel(g1[f], olv[1][f]) < 1e-4, ("C2 level 1", f, _rel(g1[f], olv[1][f]))
    assert_rows_close(g1, olv[1], "C2 level 1")
    assert np.array_equal(_np(g1["is_parent"]), olv[1]["is_parent"])
    # levels 1-3 end to end: counts within 0.1 % (a pair within 1e-7 of a gate threshold may flip after level 1) and the
    # global moments of the mixture to 1e-4 -- unconditionally
    for k in (1, 2, 3):
        ng, no = lists[0][k]["xyz"].shape[0], olv[k]["xyz"].shape[0]
        assert abs(ng - no) <= max(1, no // 1000), ("C2 count", k, ng, no)
        assert_moments_close(lists[0][k], olv[k], ("C2 end to end", k))
    # levels 2 and 3 cascade-free: the GPU level computed from the ORACLE's previous level (arrays, weights, flags)
    for k in (2, 3):
        prev = olv[k - 1]
        with hem.HemMixture(**HEM_PARAMS) as m:
            m.set_level0(prev["xyz"], prev["color"], prev["opacity"], prev["cov6"], prev["sh"])
            m.set_state(parent_mask=prev["is_parent"], weight=prev["weight"])
            _, dropped = m.run_level()
            st = m.stats()
            got = m.get_level(with_state=True)
        assert (st["parents"], st["pairs"], st["orphans"], dropped) == (ost[k - 1]["parents"], ost[k - 1]["pairs"], ost[k - 1]["orphans"], ost[k - 1]["dropped"]), ("C2 cascade-free", k)
        for f in ("xyz", "color", "cov6", "opacity", "sh", "weight"):
            assert _rel(got[f], olv[k][f]) < 1e-4, ("C2 cascade-free", k, f, _rel(got[f], olv[k][f]))
        assert_rows_close(got, olv[k], ("C2 cascade-free", k))

    # ---- point-to-plane ICP, coarse to fine, against the oracle's chain on the same level lists
    res, clouds = gpu_multiscale(lists[0], lists[1])
    T_final = res[-1][1].transformation
    assert np.linalg.norm(T_final - T_gt) < 1e-3, (T_final, T_gt)
    T = np.eye(4)
    for k, (s, t) in enumerate(clouds):
        nrm = oracle.normals_from_cov(t.covariances)
        w = oracle.icp(s.points, t.points, nrm, T, kind=1, max_corr=MAX_CORR[k], max_iter=ITER_VALUES[k])
        g = res[k][1]
        assert np.linalg.norm(g.transformation - w["transformation"]) < 1e-5, ("C2 ICP level", k, g.transformation, w["transformation"])
        assert g.iterations == w["iterations"] and abs(g.fitness - w["fitness"]) < 1e-9 and abs(g.inlier_rmse - w["inlier_rmse"]) < 1e-7
        T = w["transformation"]


def test_c3_5m_level_equals_the_reference_digest():
    """BASELINE configs[2] against the reference's VALUES at its own size: level 1 of the bench's own 5 M-splat cloud
    (synth.make_cloud(5_000_000, seed=0); mixture.cpp:66-285 through mixture_wrapper.cpp:10-18) = tests/golden/hem_5m_digest.npz --
    rows, parents, accepted pairs, orphans, dropped rows and rand() draws EXACT, the new parent flag of EVERY row exact, float64 global
    moments, per-1024-row column sums over all rows and 10 000 sampled rows (xyz, all six covariance entries, colour, opacity, SH,
    weight; each on the row's own scale) to 1e-4.  Levels 2 and 3 of the same hierarchy end to end: sizes within 0.1 % (a pair within
    1e-7 of a gate can flip behind level 1), global moments to 1e-4."""
    import digest5m
    from gaussiansplattingregistration_amd import hem, synth
    want = dict(np.load(os.path.join(GOLDEN, "hem_5m_digest.npz")))
    cloud = synth.make_cloud(5_000_000, seed=0)
    assert digest5m.input_hash(cloud) == bytes(want["input_sha256"]).decode(), "this box drew a different cloud than the fixture's (numpy version?)"
    with hem.HemMixture(rng_mode="glibc", **HEM_PARAMS) as m:
        m.set_rng("glibc", 1, 0)
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        for k in (1, 2, 3):
            _, dropped = m.run_level()
            st = m.stats()
            lv = {f: _np(v) for f, v in m.get_level(with_state=True).items()}
            if k == 1:
                got = digest5m.digest(lv, {"parents": st["parents"], "pairs": st["pairs"], "orphans": st["orphans"], "dropped": dropped,
                                           "draws": st["rng_draws"]}, idx=want["sample_idx"])
                bad = digest5m.compare(got, want, tol=1e-4)
                assert not bad, bad
            else:
                no = int(want[f"l{k}_n_out"])
                assert abs(lv["xyz"].shape[0] - no) <= max(1, no // 1000), ("C3 level size", k, lv["xyz"].shape[0], no)
                assert abs(st["pairs"] - int(want[f"l{k}_pairs"])) <= max(1, int(want[f"l{k}_pairs"]) // 1000)
                g = digest5m.global_moments(lv)
                o = {key: want[f"l{k}_g_{key}"] for key in g}
                assert abs(g["W"] - o["W"]) <= 1e-4 * o["W"]
                assert np.abs(g["mean"] - o["mean"]).max() <= 1e-4 * o["extent"]
                assert np.abs(g["cov"] - o["cov"]).max() <= 1e-4 * np.abs(o["cov"]).max()
                assert np.abs(g["color"] - o["color"]).max() <= 1e-4 * o["rms_color"]
                assert abs(g["opacity"] - o["opacity"]) <= 1e-4 * o["rms_opacity"]
                assert np.abs(g["sh"] - o["sh"]).max() <= 1e-4 * o["rms_sh"]


def _level1_against_digest(fixture, seed, shape):
    """Level 1 of synth.make_cloud(5_000_000, seed, shape) against the reference's digest `fixture` (tests/golden/make_golden_5m.py --shape ...): parents,
    pairs, orphans exact; the validity erase's count exact or -- the documented class: det <= 0 of a near-singular MERGED covariance follows the
    summation order of the M-step -- off by a handful, in which case the rows no longer line up and the comparison falls back to the global
    moments.  -> (dropped on the GPU, dropped by the reference)."""
    import digest5m
    from gaussiansplattingregistration_amd import hem, synth
    want = dict(np.load(os.path.join(GOLDEN, fixture)))
    cloud = synth.make_cloud(5_000_000, seed=seed, shape=shape)
    assert digest5m.input_hash(cloud) == bytes(want["input_sha256"]).decode(), "this box drew a different cloud than the fixture's (numpy version?)"
    with hem.HemMixture(rng_mode="glibc", **HEM_PARAMS) as m:
        m.set_rng("glibc", 1, 0)
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        _, dropped = m.run_level()
        st = m.stats()
        lv = {f: _np(v) for f, v in m.get_level(with_state=True).items()}
    assert (st["parents"], st["pairs"], st["orphans"]) == (int(want["parents"]), int(want["pairs"]), int(want["orphans"]))
    assert abs(dropped - int(want["dropped"])) <= 4, (dropped, int(want["dropped"]))
    if dropped == int(want["dropped"]):
        got = digest5m.digest(lv, {"parents": st["parents"], "pairs": st["pairs"], "orphans": st["orphans"], "dropped": dropped, "draws": st["rng_draws"]},
                              idx=want["sample_idx"])
        bad = digest5m.compare(got, want, tol=1e-4)
        assert not bad, bad
    else:                   # (rows shifted by the differing erase: the level as a whole)
        g = digest5m.global_moments(lv)
        o = {key[2:]: want[key] for key in want if key.startswith("g_")}
        assert abs(g["W"] - o["W"]) <= 1e-4 * o["W"] and np.abs(g["mean"] - o["mean"]).max() <= 1e-4 * o["extent"]
        assert np.abs(g["cov"] - o["cov"]).max() <= 1e-4 * np.abs(o["cov"]).max() and np.abs(g["sh"] - o["sh"]).max() <= 1e-4 * o["rms_sh"]
        assert np.abs(g["color"] - o["color"]).max() <= 1e-4 * o["rms_color"] and abs(g["opacity"] - o["opacity"]) <= 1e-4 * o["rms_opacity"]
    return dropped, int(want["dropped"]), st


def test_surfel_5m_level_equals_the_reference_digest():
    """The size of BASELINE configs[2] on the SURFEL recipe (what trained 3DGS scenes look like: 60 % discs, 15 % needles, covariance condition
    numbers 1e2 .. 1e5): the reference's level 1 has 3 054 641 rows, 21 of them erased (tests/golden/hem_5m_aniso_digest.npz; the oracle that
    computed it equals oracle/_ref bit for bit on the 1 M cloud of this recipe)."""
    got, want, st = _level1_against_digest("hem_5m_aniso_digest.npz", 12, "aniso")
    assert want == 21 and st["irregular"] > 0


def test_clustered_5m_level_equals_the_reference_digest():
    """... and on the large-scene recipe (README.md:113 of the reference: 60 % of the splats in 40 clumps of 30 - 100 x the background density, six giant
    splats whose search spheres span the scene, far outliers): heavy parents cut into work items, crowded sum buckets, long grid rows.  The
    reference's level 1: 1 710 339 rows, 37 263 229 pairs (tests/golden/hem_5m_clustered_digest.npz; oracle = oracle/_ref bit for bit at 300 k splats of
    this recipe)."""
    got, want, st = _level1_against_digest("hem_5m_clustered_digest.npz", 21, "clustered")
    assert st["heavy_parents"] > 0 and st["one_pass"] == 1


def test_c3_2x5m_four_level_coarse_to_fine(oracle):
    """BASELINE configs[2] -- what bench.py runs: 2 x 5 M splats (SH degree 3), 3 HEM levels per cloud on one libc rand()
    stream, 4-entry coarse-to-fine point-to-plane ICP, on bench.py's OWN pair (5 degrees about (1,1,1)/sqrt(3),
    0.05 h (1,-1,0.5) apart: the coarsest level uses its whole budget of 50 iterations).  Properties per level, distance to the
    ground-truth motion, and the oracle's ICP on ALL FOUR entries of the schedule (185 k / 556 k / 1.67 M / 5 M points; the values of the
    HEM half at this size: test_c3_5m_level_equals_the_reference_digest)."""
    import bench
    n = 5_000_000
    assert (bench.PAIR_ANGLE_DEG, bench.PAIR_SHIFT_H, bench.ITER_VALUES, bench.MAX_CORR) == (5.0, 0.05, ITER_VALUES, MAX_CORR)
    src, tgt, T_gt = make_pair_torch(n, seed=100, angle_deg=bench.PAIR_ANGLE_DEG, shift=bench.PAIR_SHIFT_H)
    lists, stats = gpu_level_lists(src, tgt)
    h = tgt["h"]
    for ci in range(2):
        for k in range(3):
            level_properties(lists[ci][k], lists[ci][k + 1], stats[ci][k], stats[ci][k]["dropped_now"], h, ("C3", ci, k))
        sizes = [lv["xyz"].shape[0] for lv in lists[ci]]
        assert sizes[0] == n and all(0.30 * a < b < 0.37 * a for a, b in zip(sizes, sizes[1:])), sizes
    # the two clouds share one rand() stream: the second cloud's flags continue where the first cloud's levels stopped
    draws = stats[1][-1]["rng_draws"]
    assert draws == sum(lv["xyz"].shape[0] + s["dropped_now"] for c in range(2) for lv, s in zip(lists[c][1:], stats[c])) + 2 * n
    # level 0 flags of the first cloud = the reference's first n draws
    assert np.array_equal(_np(lists[0][0]["is_parent"][:200000]), oracle.parent_flags(200000, 3.0))
    # the mixture keeps the scene: global mean and total covariance of every level equal level 0's (moment matching)
    for ci in range(2):
        base = global_moments(lists[ci][0])
        for k in (1, 2, 3):
            g = global_moments(lists[ci][k])
            assert abs(g["W"] - base["W"]) <= 1e-4 * base["W"]
            assert np.abs(g["mean"] - base["mean"]).max() <= 1e-4 * base["extent"]
    res, clouds = gpu_multiscale(lists[0], lists[1])
    T_final = res[-1][1].transformation
    assert np.linalg.norm(T_final - T_gt) < 1e-3, (T_final, T_gt)
    assert res[-1][1].fitness > 0.99
    assert res[0][1].iterations == ITER_VALUES[0], res[0][1].iterations          # the 5 degree pair uses the coarsest level's whole budget
    # oracle ICP on every entry (185 k, 556 k, 1.67 M and 5 M x 5 M points), chained like the driver does
    T = np.eye(4)
    for k in range(4):
        s, t = clouds[k]
        nrm = oracle.normals_from_cov(t.covariances)
        w = oracle.icp(s.points, t.points, nrm, T, kind=1, max_corr=MAX_CORR[k], max_iter=ITER_VALUES[k])
        g = res[k][1]
        assert np.linalg.norm(g.transformation - w["transformation"]) < 1e-5, ("C3 ICP level", k)
        assert g.iterations == w["iterations"]
        T = w["transformation"]


def test_anisotropic_1m_level_equals_oracle(oracle):
    """The surfel workload at 1 M splats (bench.py reports its 5 M level beside the isotropic one): level 1 of the GPU equals the
    oracle's in every discrete outcome -- parents, accepted pairs, orphans, dropped -- and per component to 1e-4; its level 2,
    computed from the ORACLE's level 1, likewise.  Most of the cloud lies inside the stage-1 filter's precondition although
    three quarters of it have covariance condition numbers beyond the 80 the round-2 predicate admitted."""
    from gaussiansplattingregistration_amd import hem, synth
    n = 1_000_000
    c = synth.make_cloud(n, seed=12, shape="aniso")
    C = c["cov6"][:50000].astype(np.float64)
    ev = np.linalg.eigvalsh(np.stack([C[:, [0, 1, 2]], C[:, [1, 3, 4]], C[:, [2, 4, 5]]], 1))
    assert (ev[:, 2] > 80 * ev[:, 0]).mean() > 0.6
    o = oracle.HemOracle(c["xyz"], c["color"], c["cov6"], c["opacity"], c["sh"])
    olv, ost = [o.level(0)], []
    for k in range(2):
        o.run_level()
        ost.append(o.stats())
        olv.append(o.level(k + 1))
    o.close()
    for k in (1, 2):
        prev = olv[k - 1]
        with hem.HemMixture(**HEM_PARAMS) as m:
            m.set_level0(prev["xyz"], prev["color"], prev["opacity"], prev["cov6"], prev["sh"])
            m.set_state(parent_mask=prev["is_parent"], weight=prev["weight"])
            _, dropped = m.run_level()
            st = m.stats()
            got = m.get_level(with_state=True)
        assert (st["parents"], st["pairs"], st["orphans"], dropped) == (ost[k - 1]["parents"], ost[k - 1]["pairs"], ost[k - 1]["orphans"], ost[k - 1]["dropped"]), ("aniso 1 M", k)
        if k == 1:
            assert st["irregular"] < 0.25 * n, st["irregular"]
        for f in ("xyz", "color", "cov6", "opacity", "sh", "weight"):
            assert _rel(got[f], olv[k][f]) < 1e-4, ("aniso 1 M", k, f, _rel(got[f], olv[k][f]))
        assert_rows_close(got, olv[k], ("aniso 1 M", k))


def test_clustered_1m_level_equals_oracle(oracle):
    """The large-scene shape (README.md:113 of the reference: "for larger scenes the HEM downsampler becomes extremely slow"): 60 % of
    a 1 M-splat cloud in 40 clumps of 30-100 x the background density, giant background splats whose search spheres span the scene
    (the reference's grid cell is the largest parent radius, mixture.cpp:92-99), far outliers.  Level 1 of the GPU equals the
    oracle's in every discrete outcome and per component to 1e-4; the level's fallback flags are looked at, not assumed."""
    from gaussiansplattingregistration_amd import hem, synth
    n = 1_000_000
    c = synth.make_cloud(n, seed=21, shape="clustered")
    want, wst = oracle.hem(c, 1)
    with hem.HemMixture(**HEM_PARAMS) as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        _, dropped = m.run_level()
        st = m.stats()
        got = m.get_level(with_state=True)
    assert (st["parents"], st["pairs"], st["orphans"], dropped) == (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"]), (st, wst[0])
    assert st["heavy_parents"] > 0 and st["heavy_work_items"] >= st["heavy_parents"] and st["one_pass"] == 1
    for f in ("xyz", "color", "cov6", "opacity", "sh"):
        assert _rel(got[f], want[0][f]) < 1e-4, ("clustered 1 M", f, _rel(got[f], want[0][f]))
    # per component on its own scale -- the far outliers (orphans, copied bit for bit) and the giants included
    assert_rows_close(got, want[0], "clustered 1 M")
