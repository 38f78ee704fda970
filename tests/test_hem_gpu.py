"""HEM parity proper: the HIP path, called through the C ABI (via the mixture_bind-shaped front end),
against the reference's golden vectors and against the oracle on seeded inputs.

Tolerance (BASELINE.json north_star): mixture moments within 1e-4 relative.  The discrete decisions
(radius / colour / KL gates, parent rule, orphan test) are bit-exact by construction (gsr_math.h), so
component COUNTS must be equal; float sums differ from the reference only by summation order."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, HEM_CASES, golden_cloud, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def _check_level_mostly(got, want, tag, frac=0.995):
    """Levels >= 2 of an end-to-end run: rounding of level k can flip a pair that sits within 1e-7 of a gate
    threshold at level k+1, which moves ONE component by ~1e-3.  Equal counts and
    per-component agreement of all but a few components are required."""
    assert got["xyz"].shape == want["xyz"].shape, (tag, got["xyz"].shape, want["xyz"].shape)
    scale = np.abs(want["xyz"]).max() + 1e-30
    ok = np.abs(got["xyz"].astype(np.float64) - want["xyz"]).max(axis=1) / scale < TOL
    assert ok.mean() >= frac, (tag, float(ok.mean()))


def _check_level(got, want, tag):
    assert got["xyz"].shape == want["xyz"].shape, (tag, got["xyz"].shape, want["xyz"].shape)
    for f in ("xyz", "color", "cov6", "opacity", "sh"):
        assert _rel(got[f], want[f]) < TOL, (tag, f, _rel(got[f], want[f]))


@pytest.mark.parametrize("case", HEM_CASES)
def test_gpu_vs_reference_golden(case):
    from gaussiansplattingregistration_amd import mixture_bind as mb
    g = load_golden(case)
    L = int(g["levels"])
    mb.reset_rng(position=int(g["pre_draws"]))
    lv0 = mb.MixtureLevel.CreateMixtureLevel(g["xyz"].tolist() if case == "hem_deg0" else g["xyz"], g["color"], g["opacity"],
                                             g["cov6"], g["sh"])
    levels = mb.MixtureCreator.CreateMixture(L, float(g["rho"]), float(g["delta"]), float(g["kappa"]), float(g["tau"]), lv0)
    assert len(levels) == L                                    # level 0 removed (mixture_wrapper.cpp:14-17)
    for k in range(L):
        xyz, col, op, cov, sh = mb.MixtureLevel.CreateArrays(levels[k])
        got = {"xyz": xyz, "color": col, "opacity": op, "cov6": cov, "sh": sh}
        want = {f: g[f"out_{f}_{k}"] for f in got}
        _check_level(got, want, (case, k))
    if case == "hem_noparent":                                 # unchanged level is a bit-exact copy
        assert np.array_equal(levels[0].pointSet, g["xyz"]) and np.array_equal(levels[0].features, g["sh"])


def test_device_tensors_zero_copy_path_equals_host_path():
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(5000, seed=4, h=0.7)
    host, _ = hem.create_mixture(c, 2)
    dev_in = {k: torch.from_numpy(v).cuda() for k, v in c.items() if isinstance(v, np.ndarray)}
    dev, _ = hem.create_mixture(dev_in, 2, as_torch=True)
    for k in range(2):
        for f in ("xyz", "color", "cov6", "opacity", "sh"):
            assert np.array_equal(host[k][f], dev[k][f].cpu().numpy()), (k, f)     # deterministic: bit-identical


def test_known_answer_50k_level_counts_and_pairs(oracle):
    """SURVEY 8(c) known answer from the reference: 50 000 splats, box +-1.5, seed 0 -> 16498 / 5488 / 1852."""
    from gaussiansplattingregistration_amd import hem, synth
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["counts_only"]["n50000_h1.5_seed0"]
    c = synth.make_cloud(50000, seed=0, h=1.5)
    got, st = hem.create_mixture(c, 3)
    assert [g["xyz"].shape[0] for g in got] == ka
    want, wst = oracle.hem(c, 1)
    # level 1 decisions are bit-exact: same parents, same accepted pairs, same orphans
    assert (st[0]["parents"], st[0]["pairs"], st[0]["orphans"]) == (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"])
    _check_level(got[0], want[0], "50k level 1")


def test_single_level_cascade_free_at_200k(oracle):
    """Level k+1 fed with the ORACLE's level k (arrays, weights, parent flags): a one-level comparison that
    does not cascade earlier rounding into later discrete decisions."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(200000, seed=1)
    o = oracle.HemOracle(c["xyz"], c["color"], c["cov6"], c["opacity"], c["sh"])
    o.run_level()
    l1 = o.level(1)
    o.run_level()
    l2, st2 = o.level(2), o.stats()
    o.close()
    with hem.HemMixture() as m:
        m.set_level0(l1["xyz"], l1["color"], l1["opacity"], l1["cov6"], l1["sh"])
        m.set_state(parent_mask=l1["is_parent"], weight=l1["weight"])
        n_out, dropped = m.run_level()
        st = m.stats()
        got = m.get_level(with_state=True)
    assert (st["parents"], st["pairs"], st["orphans"], dropped) == (st2["parents"], st2["pairs"], st2["orphans"], st2["dropped"])
    _check_level(got, l2, "200k level 2 from oracle level 1")
    assert _rel(got["weight"], l2["weight"]) < TOL


def test_global_moments_multi_level(oracle):
    """End-to-end 3 levels at 100k: per-level counts within 0.1% and the FULL global moments of the mixture within 1e-4
    (sum of weights, weighted mean, weighted total covariance, mean colour / opacity / SH) -- asserted unconditionally,
    also when a count differs by one."""
    from gaussiansplattingregistration_amd import hem, synth
    from test_configs_gpu import assert_moments_close
    c = synth.make_cloud(100000, seed=2)
    want, _ = oracle.hem(c, 3)
    with hem.HemMixture() as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        for k in range(3):
            m.run_level()
            got = m.get_level(with_state=True)
            nw = want[k]["xyz"].shape[0]
            assert abs(got["xyz"].shape[0] - nw) <= max(1, nw // 1000), (k, got["xyz"].shape[0], nw)
            assert abs(got["weight"].astype(np.float64).sum() - 100000.0) < 1e-4 * 100000.0      # weight conserved through levels
            assert_moments_close(got, want[k], ("100k end to end", k))


def test_full_size_properties_5m():
    """BASELINE full size (5 M splats, SH degree 3): size-independent properties of one level."""
    from gaussiansplattingregistration_amd import hem, synth
    n = 5_000_000
    c = synth.make_cloud_torch(n, seed=0)
    xyz, col, op, cov6, sh, h = c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], c["h"]
    with hem.HemMixture(rng_mode="hash", rng_seed=7) as m:
        m.set_level0(xyz, col, op, cov6, sh)
        l0 = m.get_level(as_torch=True, with_state=True)
        n_out, dropped = m.run_level()
        st = m.stats()
        l1 = m.get_level(as_torch=True, with_state=True)
    assert dropped == 0 and n_out == st["parents"] + st["orphans"]
    assert abs(int(l0["is_parent"].sum()) - n / 3) < 0.01 * n                     # parent probability 1/rho
    assert st["parents"] == int(l0["is_parent"].sum())
    # responsibilities of every claimed child sum to one -> total weight conserved
    assert abs(float(l1["weight"].double().sum()) - n) < 1e-4 * n
    # orphans are bit-exact copies of inputs (checked through a checksum of their rows)
    P = st["parents"]
    orph = l1["xyz"][P:]
    assert orph.shape[0] == st["orphans"]
    # weighted mean of the mixture is preserved by moment matching
    m0 = xyz.double().mean(0)
    m1 = (l1["weight"].double()[:, None] * l1["xyz"].double()).sum(0) / l1["weight"].double().sum()
    assert float((m0 - m1).abs().max()) < 1e-4 * h
    # covariances stay positive definite, nothing non-finite
    assert bool(torch.isfinite(l1["xyz"]).all()) and bool(torch.isfinite(l1["sh"]).all())
    c = l1["cov6"].double()
    det = (-c[:, 2] * c[:, 2] * c[:, 3] + 2 * c[:, 1] * c[:, 2] * c[:, 4] - c[:, 0] * c[:, 4] * c[:, 4]
           - c[:, 1] * c[:, 1] * c[:, 5] + c[:, 0] * c[:, 3] * c[:, 5])
    assert bool((det > 0).all())


@pytest.mark.gpu
def test_large_cloud_30m_properties():
    """30 M splats, one level: past 2^24 components and 2^32 scanned candidates (64-bit offsets everywhere), the same
    size-independent properties as the 5 M test.  (SH degree 1 keeps the footprint at ~40 GB.)"""
    import torch
    from gaussiansplattingregistration_amd import hem, synth
    n = 30_000_000
    c = synth.make_cloud_torch(n, seed=3, sh_degree=1)
    with hem.HemMixture(rng_mode="hash", rng_seed=11) as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        n_out, dropped = m.run_level()
        st = m.stats()
        l1 = m.get_level(as_torch=True, with_state=True)
    assert st["candidates"] > 2 ** 32 and st["pairs"] > 2 ** 28
    assert dropped == 0 and n_out == st["parents"] + st["orphans"] and abs(st["parents"] - n / 3) < 0.01 * n
    assert abs(float(l1["weight"].double().sum()) - n) < 1e-4 * n                 # total weight conserved
    m0 = c["xyz"].double().mean(0)
    m1 = (l1["weight"].double()[:, None] * l1["xyz"].double()).sum(0) / l1["weight"].double().sum()
    assert float((m0 - m1).abs().max()) < 1e-4 * c["h"]                            # weighted mean preserved
    assert bool(torch.isfinite(l1["xyz"]).all()) and bool(torch.isfinite(l1["sh"]).all()) and bool(torch.isfinite(l1["cov6"]).all())


def test_device_logf_and_kld_bit_exact_vs_host_libm(hip_lib, oracle):
    """The KL gate is bit-exact only if the device logf equals libm's: check it on 2M inputs, plus KLD itself."""
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 1_000_000)), rng.uniform(0.5, 2.0, 1_000_000),
                        [0.0, 1.0, np.inf, 1e-45, 3e-39, -1.0, np.nan]]).astype(np.float32)
    out = np.empty_like(x)
    assert hip_lib.gsr_debug_logf(x.ctypes.data, x.size, out.ctypes.data, 0) == 0
    with np.errstate(all="ignore"):
        want = np.log(x)                  # numpy float32 log may differ from libm: use the oracle's libm call instead
    want = np.array([oracle.logf(float(v)) for v in x[:20000]], np.float32)
    a, b = out[:20000].view(np.uint32), want.view(np.uint32)
    nan = np.isnan(out[:20000]) & np.isnan(want)
    assert np.array_equal(a[~nan], b[~nan])
    assert np.isnan(out[-1]) and np.isnan(out[-2]) and out[-5] == np.inf and out[-7] == -np.inf
    from gaussiansplattingregistration_amd import synth
    a_, b_ = synth.make_cloud(100000, seed=5), synth.make_cloud(100000, seed=6)
    pm = (a_["xyz"] + 0.1 * b_["xyz"]).astype(np.float32)
    got = np.empty(100000, np.float32)
    assert hip_lib.gsr_debug_kld(a_["xyz"].ctypes.data, a_["cov6"].ctypes.data, pm.ctypes.data, b_["cov6"].ctypes.data,
                                 100000, got.ctypes.data, 0) == 0
    want = oracle.kld(a_["xyz"], a_["cov6"], pm, b_["cov6"])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_empty_and_tiny_inputs():
    from gaussiansplattingregistration_amd import hem
    z = np.zeros
    with hem.HemMixture() as m:
        m.set_level0(z((0, 3), np.float32), z((0, 3), np.float32), z((0,), np.float32), z((0, 6), np.float32), z((0, 0), np.float32))
        assert m.run_level() == (0, 0)
    with hem.HemMixture(hem_reduction=1.0) as m:        # one splat, certainly a parent: merges with itself
        m.set_level0(np.float32([[1, 2, 3]]), np.float32([[0.1, 0.2, 0.3]]), np.float32([0.5]),
                     np.float32([[0.01, 0, 0, 0.02, 0, 0.03]]), np.float32([[0.5] * 9]))
        assert m.run_level() == (1, 0)
        lv = m.get_level()
        assert np.allclose(lv["xyz"], [[1, 2, 3]]) and np.allclose(lv["cov6"], [[0.01, 0, 0, 0.02, 0, 0.03]], rtol=1e-6)


def test_hash_rng_mode_statistics():
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(60000, seed=8)
    with hem.HemMixture(rng_mode="hash", rng_seed=123) as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        f = m.get_level(with_state=True)["is_parent"]
        assert abs(f.mean() - 1 / 3) < 0.01
        n1, _ = m.run_level()
        assert 0.30 * 60000 < n1 < 0.37 * 60000


def test_outliers_clusters_and_duplicates(oracle):
    """Non-uniform input: two dense clusters, far-away outliers (the grid is laid over a robust box and the
    outliers are clamped into boundary cells), exact duplicates, a few giant splats."""
    from gaussiansplattingregistration_amd import hem, synth
    rng = np.random.default_rng(5)
    c = synth.make_cloud(30000, seed=21, h=1.2, sh_degree=1)
    xyz = c["xyz"]
    xyz[:12000] = xyz[:12000] * 0.25 + np.float32([0.6, 0.6, 0.6])          # dense cluster
    xyz[12000:20000] = xyz[12000:20000] * 0.15 - np.float32([0.8, 0.2, 0.5])  # denser cluster
    out_idx = rng.choice(30000, 40, replace=False)
    xyz[out_idx] = (rng.normal(size=(40, 3)) * 500.0).astype(np.float32)     # outliers 400x the scene extent
    xyz[100:110] = xyz[90:100]                                               # exact duplicates
    c["cov6"][out_idx[:5]] *= np.float32(2.0e5)                              # giant splats among the outliers
    c["cov6"][500:505] *= np.float32(400.0)                                  # giant splats inside
    want, wst = oracle.hem(c, 2)
    with hem.HemMixture() as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        for k in range(2):
            m.run_level()
            st = m.stats()
            got = m.get_level()
            if k == 0:
                assert (st["parents"], st["pairs"], st["orphans"], st["dropped"]) == (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"])
                _check_level(got, want[k], ("outliers", k))
            else:
                _check_level_mostly(got, want[k], ("outliers", k))


def test_two_pass_fallback_equals_sparse_path(monkeypatch):
    """GSR_HEM_SPARSE_GB=0 forces the COUNT + FILL fallback; it must give the same pairs as the one-pass path."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(20000, seed=31, h=1.1, sh_degree=1)
    a, _ = hem.create_mixture(c, 2)
    monkeypatch.setenv("GSR_HEM_SPARSE_GB", "0")
    b, _ = hem.create_mixture(c, 2)
    for k in range(2):
        assert a[k]["xyz"].shape == b[k]["xyz"].shape
        for f in ("xyz", "cov6", "sh", "opacity"):
            assert _rel(a[k][f], b[k][f]) < 1e-5, (k, f)


def test_heavy_parent_work_items_change_nothing(monkeypatch):
    """Heavy parents are cut into work items of 8192 candidates served from a queue (GSR_HEM_SPLIT=0: one wave per parent).
    The parts are concatenated in order, so the pair list -- and with it every sum of the level -- is the same bit for bit,
    on the one-pass path and on the COUNT + FILL fallback."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(200000, seed=35, sh_degree=1)
    monkeypatch.setenv("GSR_HEM_SPLIT", "0")
    ref, rst = hem.create_mixture(c, 2)
    assert max(s["candidates"] for s in rst) > 0
    for budget in (None, "0"):
        monkeypatch.setenv("GSR_HEM_SPLIT", "1")
        if budget is not None:
            monkeypatch.setenv("GSR_HEM_SPARSE_GB", budget)
        got, st = hem.create_mixture(c, 2)
        for k in range(2):
            assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"])
            for f in ("xyz", "color", "cov6", "sh", "opacity"):
                assert np.array_equal(got[k][f], ref[k][f]), (budget, k, f)


@pytest.mark.parametrize("shape", ["iso", "aniso", "clustered"])
def test_parents_per_selection_wave_change_nothing(monkeypatch, shape):
    """A selection wave takes up to four consecutive parents of the processing order and keeps its survivor ring and its third-stage
    queue across them (k_select, SEL_NP); the rings are FIFO, so every parent's pairs come out in the order -- and at the places -- one
    parent per wave wrote them: GSR_HEM_SELECT_NP = 1, 2, 4 give the same levels bit for bit, on the one-pass path and on the COUNT +
    FILL fallback.  The clouds hold irregular components (pass B), heavy parents (the queue kernel beside the light one) and parents
    with several batches of survivors (batches that mix the entries of two or three parents)."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(250000, seed=77, sh_degree=1, shape=shape)
    c["cov6"][5::997] = np.array([1.0, 0, 0, 1.0, 0, -1.0], np.float32)       # irregular: not positive definite
    monkeypatch.setenv("GSR_HEM_SELECT_NP", "1")
    ref, rst = hem.create_mixture(c, 2)
    assert rst[0]["irregular"] > 0 and rst[0]["heavy_parents"] > 0, rst[0]
    for np_, budget in (("2", None), ("4", None), ("4", "0"), ("1", "0")):
        monkeypatch.setenv("GSR_HEM_SELECT_NP", np_)
        if budget is None:
            monkeypatch.delenv("GSR_HEM_SPARSE_GB", raising=False)
        else:
            monkeypatch.setenv("GSR_HEM_SPARSE_GB", budget)
        got, st = hem.create_mixture(c, 2)
        for k in range(2):
            assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"], rst[k]["dropped"]), (np_, budget, k)
            for f in ("xyz", "color", "cov6", "sh", "opacity"):
                assert np.array_equal(got[k][f].view(np.uint32), ref[k][f].view(np.uint32)), (np_, budget, k, f)


def _levels_on_one_context(m, c, levels, zero_copy):
    out, stats = [], []
    m.set_rng("glibc", 1, 0)
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
    for _ in range(levels):
        m.run_level(out=m.new_output() if zero_copy else None)
        stats.append(m.stats())
        lv = m.get_level(as_torch=zero_copy, with_state=True)
        out.append({k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in lv.items()})
    return out, stats


@pytest.mark.parametrize("shape,bad_every", [("iso", 997), ("aniso", 997), ("clustered", 997), ("iso", 9973), ("aniso", 19997)])
def test_asynchronous_level_equals_the_synchronous_one(monkeypatch, shape, bad_every):
    """A level without a host round trip between its first and its last kernel (the default once the context's buffers exist: sizes stay on
    the device, launches cover bounds, every write is clamped, ONE answer comes back with the next level's prologue) against the
    synchronous schedule of rounds 1-4 (GSR_HEM_ASYNC=0: candidates, pairs, orphans and surviving rows read back on the way).  Three
    levels, bit for bit, own buffers and zero-copy output, on a cloud with irregular components, heavy parents and erased rows; the
    statistics say which schedule ran and how many round trips it took."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(120000, seed=31, sh_degree=2, shape=shape)
    # irregular: not positive definite -> erased from level 1.  Every 997th: 121 rows, more than the device erases in place (ERASE_MAX = 32: the
    # host's scan + compaction, a second round trip); every 9973rd / 19997th: a handful, erased in place inside the asynchronous level
    c["cov6"][5::bad_every] = np.array([1.0, 0, 0, 1.0, 0, -1.0], np.float32)
    monkeypatch.setenv("GSR_HEM_ASYNC", "0")
    with hem.HemMixture() as m:
        ref, rst = _levels_on_one_context(m, c, 3, False)
    assert all(s["schedule"] == 0 and s["round_trips"] >= 4 for s in rst), [(s["schedule"], s["round_trips"]) for s in rst]
    assert rst[0]["dropped"] > 0
    monkeypatch.delenv("GSR_HEM_ASYNC")
    for zero_copy in (False, True):
        with hem.HemMixture() as m:
            first, fst = _levels_on_one_context(m, c, 3, zero_copy)           # a fresh context: its first level sizes the buffers synchronously
            again, ast = _levels_on_one_context(m, c, 3, zero_copy)           # ... and from then on every level is asynchronous
        assert fst[0]["schedule"] == 0 and all(s["schedule"] == 1 for s in fst[1:]), [s["schedule"] for s in fst]
        assert all(s["schedule"] == 1 for s in ast), [s["schedule"] for s in ast]
        # one round trip per level -- two when more rows are erased than the device does in place (the host finishes the compaction and
        # asks for the next prologue again)
        assert all(s["round_trips"] == (2 if s["dropped"] > 32 else 1) for s in ast), [(s["round_trips"], s["dropped"]) for s in ast]
        for got, st in ((first, fst), (again, ast)):
            for k in range(3):
                for f in ("parents", "pairs", "orphans", "dropped", "candidates", "n_out", "heavy_parents", "rng_draws", "max_pairs_of_a_parent"):
                    assert st[k][f] == rst[k][f], (zero_copy, k, f, st[k][f], rst[k][f])
                for f in ("xyz", "color", "cov6", "sh", "opacity", "weight", "is_parent"):
                    assert np.array_equal(got[k][f].view(np.uint8), ref[k][f].view(np.uint8)), (zero_copy, k, f)


def test_repeated_surfel_level_erases_on_the_device_in_one_round_trip():
    """the buffer of the device-side validity erase was first allocated by the host's path for n_pre rows, the
    asynchronous check asked for the level's INPUT size (1.6 x n_pre on a surfel level, 3 x on an isotropic one), so every later level found
    it too small, only counted the erased rows and took the host's path again -- a second erase, a second prologue, a second round trip,
    and k_erase_save / k_erase_shift fed by the device count never ran on a real level.  A surfel cloud whose level 1 drops a merged
    component (oracle: 1 row at 300 k splats, seed 13), the level repeated on one context: from the second repetition on ONE round trip,
    the erase on the device, the same bits as the first (synchronous) level."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(300000, seed=13, shape="aniso", sh_degree=1)
    for zero_copy in (False, True):
        with hem.HemMixture() as m:
            runs = [_levels_on_one_context(m, c, 1, zero_copy) for _ in range(3)]
        (ref, rst), rest = runs[0], runs[1:]
        assert rst[0]["dropped"] >= 1 and rst[0]["schedule"] == 0, (rst[0]["dropped"], rst[0]["schedule"])
        for got, st in rest:
            assert st[0]["schedule"] == 1 and st[0]["dropped"] == rst[0]["dropped"] and st[0]["n_out"] == rst[0]["n_out"]
            for f in ("xyz", "color", "cov6", "sh", "opacity", "weight", "is_parent"):
                assert np.array_equal(got[0][f].view(np.uint8), ref[0][f].view(np.uint8)), (zero_copy, f)
        assert rest[-1][1][0]["round_trips"] == 1, [r[1][0]["round_trips"] for r in rest]


@pytest.mark.parametrize("shape", ["iso", "aniso"])
def test_all_levels_in_one_call_equal_the_level_by_level_path(shape):
    """gsr_hem_run_levels (MixtureCreator::CreateMixture in one library call, mixture_wrapper.cpp:10-18): three levels written one behind the
    other into caller-owned arenas = set_output + run_level per level, bit for bit, statistics included; the normals that leave with the
    levels (side stream) = gsr_normals_from_cov on the same rows, bit for bit, level 0's too; the context goes on from the last level (whose
    trailing prologue was skipped: the next call takes it again); an arena that is too small fails cleanly before anything is written."""
    from gaussiansplattingregistration_amd import hem, icp, synth
    c = synth.make_cloud(150000, seed=17, sh_degree=2, shape=shape)
    c["cov6"][7::19997] = np.array([1.0, 0, 0, 1.0, 0, -1.0], np.float32)           # a few rows the validity erase drops in place
    with hem.HemMixture() as m:
        ref, rst = _levels_on_one_context(m, c, 4, True)
    for rep in range(2):                    # second repetition: every level asynchronous
        with hem.HemMixture() as m:
            if rep:
                _levels_on_one_context(m, c, 3, True)
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            n0 = m.size
            arena = m.new_arena(int(1.5 * n0) + 256, normals=True)
            nrm0 = torch.empty((n0, 3), dtype=torch.float64, device="cuda")
            levels, st = m.run_levels(3, arena=arena, normals0=nrm0)
            assert [s["n_out"] for s in st] == [s["n_out"] for s in rst[:3]]
            offs = [int(lv["xyz"].data_ptr() - arena["xyz"].data_ptr()) // 12 for lv in levels]
            assert offs[0] == 0 and all(o % 64 == 0 for o in offs) and all(b >= a + lv["xyz"].shape[0] for a, b, lv in zip(offs, offs[1:], levels))
            for k in range(3):
                for f in ("parents", "pairs", "orphans", "dropped", "candidates", "n_in", "n_out", "rng_draws", "heavy_parents", "max_pairs_of_a_parent"):
                    assert st[k][f] == rst[k][f], (rep, k, f, st[k][f], rst[k][f])
                assert st[k]["dropped_now"] == rst[k]["dropped"]
                for f in ("xyz", "color", "cov6", "sh", "opacity"):
                    assert np.array_equal(levels[k][f].cpu().numpy().view(np.uint8), ref[k][f].view(np.uint8)), (rep, k, f)
                want = icp.normals_from_cov(levels[k]["cov6"].contiguous(), device=0)
                assert torch.equal(levels[k]["normals"], want if isinstance(want, torch.Tensor) else torch.from_numpy(want).cuda()), (rep, k)
            want0 = icp.normals_from_cov(torch.from_numpy(c["cov6"]).cuda(), device=0)
            assert torch.equal(nrm0, want0 if isinstance(want0, torch.Tensor) else torch.from_numpy(want0).cuda())
            # the context goes on: level 4 through the level-by-level entry (state and flags of level 3 are the context's)
            m.run_level()
            got4 = m.get_level(with_state=True)
            for f in ("xyz", "color", "cov6", "sh", "opacity", "weight", "is_parent"):
                assert np.array_equal(np.asarray(got4[f]).view(np.uint8), ref[3][f].view(np.uint8)), (rep, "level 4", f)
    with hem.HemMixture() as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        with pytest.raises(RuntimeError, match="arenas that hold"):
            m.run_levels(2, arena=m.new_arena(m.size + 64))                  # level 1 fits, level 2 (input ~ n/3 rows behind n/3 rows) fits too ...
            m.run_levels(1, arena=m.new_arena(10))                           # ... this one cannot
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        with pytest.raises(RuntimeError, match="arenas that hold"):
            m.run_levels(1, arena=m.new_arena(m.size - 1))


@pytest.mark.parametrize("knob", ["GSR_HEM_ASYNC=0", "GSR_HEM_SH_DIRECT=1", "GSR_HEM_SPLIT=0", "GSR_HEM_TIMING=0", "GSR_HEM_SUMLW=sort"])
def test_all_levels_in_one_call_under_the_knobs(monkeypatch, knob):
    """gsr_hem_run_levels under the library's knobs -- the synchronous schedule (whose last level skips the trailing prologue through another branch),
    SH rows read where the level lies, unsplit heavy parents, no events, sorted sums: the default's bits, and the context goes on afterwards."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(90000, seed=29, sh_degree=1, shape="clustered")
    with hem.HemMixture() as m:
        ref, rst = _levels_on_one_context(m, c, 3, True)
    k, v = knob.split("=")
    monkeypatch.setenv(k, v)
    with hem.HemMixture() as m:
        for rep in range(2):
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            levels, st = m.run_levels(2)
            for q in range(2):
                assert st[q]["n_out"] == rst[q]["n_out"] and st[q]["pairs"] == rst[q]["pairs"], (knob, rep, q)
                for f in ("xyz", "color", "cov6", "sh", "opacity"):
                    if knob == "GSR_HEM_SUMLW=sort":        # (another summation of the per-child sums: equal to 1e-5, not bit for bit)
                        assert np.allclose(levels[q][f].cpu().numpy(), ref[q][f], rtol=1e-4, atol=1e-6), (knob, q, f)
                    else:
                        assert np.array_equal(levels[q][f].cpu().numpy().view(np.uint8), ref[q][f].view(np.uint8)), (knob, rep, q, f)
            m.run_level()                                   # level 3 through the level-by-level entry: the prologue the last level skipped is taken here
            assert m.size == rst[2]["n_out"]


def test_all_levels_in_one_call_edge_cases():
    """gsr_hem_run_levels on the cases the level-by-level entry handles: no SH block (F = 0), a hierarchy without parents (rho = 1e9: every level
    returns its input, mixture.cpp:250-253), rho = 1 (every component a parent), zero levels, a tiny cloud -- each equal to run_level per level."""
    from gaussiansplattingregistration_amd import hem, synth
    cases = [("F=0", dict(n=5000, sh_degree=0), dict()), ("no parents", dict(n=3000, sh_degree=1), dict(hem_reduction=1e9)),
             ("all parents", dict(n=3000, sh_degree=1), dict(hem_reduction=1.0)), ("tiny", dict(n=7, sh_degree=1), dict())]
    for tag, ck, hk in cases:
        c = synth.make_cloud(ck["n"], seed=23, sh_degree=ck["sh_degree"], h=0.5)
        with hem.HemMixture(**hk) as m:
            ref, rst = _levels_on_one_context(m, c, 3, False)
        with hem.HemMixture(**hk) as m:
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            lv0, st0 = m.run_levels(0)
            assert lv0 == [] and st0 == [] and m.size == ck["n"], tag
            levels, st = m.run_levels(3, arena=m.new_arena(3 * ck["n"] + 256, normals=True))
            for k in range(3):
                assert st[k]["n_out"] == rst[k]["n_out"] and st[k]["rng_draws"] == rst[k]["rng_draws"], (tag, k)
                for f in ("xyz", "color", "cov6", "sh", "opacity"):
                    assert np.array_equal(levels[k][f].cpu().numpy().view(np.uint8), ref[k][f].view(np.uint8)), (tag, k, f)
                assert levels[k]["normals"].shape == (st[k]["n_out"], 3) and bool(torch.isfinite(levels[k]["normals"]).all()), (tag, k)
            got = m.get_level(with_state=True)
            for f in ("weight", "is_parent"):
                assert np.array_equal(np.asarray(got[f]).view(np.uint8), ref[2][f].view(np.uint8)), (tag, f)


def test_asynchronous_level_on_buffers_too_small_reruns_synchronously():
    """An asynchronous level runs on the buffers the context has; a level that needs more (here: a context warmed on a small cloud, then a
    cloud eight times as large) finds that out ON THE DEVICE -- clamped writes, an abort flag in its one answer -- and is run again the
    synchronous way: same result as a fresh context, schedule 2 in the statistics, and asynchronous again afterwards."""
    from gaussiansplattingregistration_amd import hem, synth
    small, big = synth.make_cloud(20000, seed=41, sh_degree=1), synth.make_cloud(160000, seed=42, sh_degree=1)
    with hem.HemMixture() as m:
        ref, rst = _levels_on_one_context(m, big, 2, False)
    with hem.HemMixture() as m:
        _levels_on_one_context(m, small, 2, False)
        got, st = _levels_on_one_context(m, big, 2, False)
        again, ast = _levels_on_one_context(m, big, 2, False)
    assert st[0]["schedule"] == 2, st[0]["schedule"]
    assert all(s["schedule"] == 1 for s in ast), [s["schedule"] for s in ast]
    for res in (got, again):
        for k in range(2):
            for f in ("xyz", "color", "cov6", "sh", "opacity", "weight", "is_parent"):
                assert np.array_equal(res[k][f].view(np.uint8), ref[k][f].view(np.uint8)), (k, f)


def test_survivor_ring_across_isolated_parents(monkeypatch, oracle):
    """The survivor ring of a selection wave holds 64 + one group of chunks (SEL_QCAP = 256) and is kept across the wave's four parents.
    ADVICE r04: a parent WITHOUT a candidate row never reached the scan's drain, so a wave could take the ring past its capacity -- parent
    A leaves 63 survivors (itself + 62 duplicates), the isolated parents B and C add their own entries (64, 65), and the first group of
    parent D (200 duplicates on top of it: 192 stage-1 survivors at once, + itself) wrapped onto A's oldest entries: accepted pairs lost,
    silently.  The four parents sit on a line (their processing order is the Morton order of their cells) with nothing else around, the
    parent flags are set by hand; the level must equal the oracle's, and one parent per wave must give the same bits as four."""
    from gaussiansplattingregistration_amd import hem
    groups = [(0.0, 62), (10.0, 0), (20.0, 0), (30.0, 200)]            # (x of the parent, duplicates at its position)
    xyz, par = [], []
    for x, dup in groups:
        xyz += [[x, 0.0, 0.0]] * (1 + dup)
        par += [1] + [0] * dup
    n = len(xyz)
    rng = np.random.default_rng(5)
    cloud = {"xyz": np.asarray(xyz, np.float32), "color": np.zeros((n, 3), np.float32), "opacity": rng.normal(0.5, 0.1, n).astype(np.float32),
             "cov6": np.tile(np.array([1e-4, 0, 0, 1e-4, 0, 1e-4], np.float32), (n, 1)), "sh": rng.normal(0, 0.1, (n, 9)).astype(np.float32)}
    mask = np.asarray(par, np.uint8)
    o = oracle.HemOracle(cloud["xyz"], cloud["color"], cloud["cov6"], cloud["opacity"], cloud["sh"])
    o.set_parent_mask(mask)
    o.run_level()
    want, wst = o.level(1), o.stats()
    o.close()
    assert (wst["parents"], wst["pairs"], wst["orphans"]) == (4, 63 + 1 + 1 + 201, 0)
    res = {}
    for np_ in ("1", "4"):
        monkeypatch.setenv("GSR_HEM_SELECT_NP", np_)
        with hem.HemMixture() as m:
            m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
            m.set_state(parent_mask=mask)
            m.run_level()
            st, got = m.stats(), m.get_level(with_state=True)
        assert (st["parents"], st["pairs"], st["orphans"], st["dropped"]) == (wst["parents"], wst["pairs"], wst["orphans"], wst["dropped"]), (np_, st)
        _check_level(got, want, ("ring across isolated parents", np_))
        assert _rel(got["weight"], want["weight"]) < TOL
        res[np_] = got
    for f in ("xyz", "color", "cov6", "sh", "opacity", "weight"):
        assert np.array_equal(res["1"][f].view(np.uint32), res["4"][f].view(np.uint32)), f


@pytest.mark.parametrize("shape", ["iso", "aniso", "clustered"])
def test_row_lists_from_the_capacity_pass_change_nothing(monkeypatch, shape):
    """``k_spans`` computes every row span of every parent for the capacities; it now also leaves the non-empty ones (up to 16 per
    pass) in scan order for ``k_select``, which reads them instead of computing them again.  GSR_HEM_ROWLIST=0 (every span computed
    twice, as before): the same levels bit for bit -- with irregular components (pass B has its own list), heavy parents (more rows than
    a list holds: they recompute), on the COUNT + FILL fallback, and when the lists would exceed GSR_HEM_ROWLIST_MAX_MB (a level of
    more than 16 M parents by default: 256 bytes per parent are kept by the context)."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(200000, seed=78, sh_degree=1, shape=shape)
    c["cov6"][7::499] = np.array([1.0, 0, 0, 1.0, 0, -1.0], np.float32)       # irregular: not positive definite
    monkeypatch.setenv("GSR_HEM_ROWLIST", "0")
    ref, rst = hem.create_mixture(c, 2)
    assert rst[0]["irregular"] > 0 and rst[0]["heavy_parents"] > 0, rst[0]
    for budget in (None, "0", "cap"):
        monkeypatch.setenv("GSR_HEM_ROWLIST", "1")
        if budget == "cap":                 # lists larger than the cap (256 bytes per parent): the level runs without them
            monkeypatch.delenv("GSR_HEM_SPARSE_GB")
            monkeypatch.setenv("GSR_HEM_ROWLIST_MAX_MB", "1")
        elif budget is not None:
            monkeypatch.setenv("GSR_HEM_SPARSE_GB", budget)
        got, st = hem.create_mixture(c, 2)
        for k in range(2):
            assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"], st[k]["candidates"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"], rst[k]["dropped"], rst[k]["candidates"]), (budget, k)
            for f in ("xyz", "color", "cov6", "sh", "opacity"):
                assert np.array_equal(got[k][f].view(np.uint32), ref[k][f].view(np.uint32)), (budget, k, f)


@pytest.mark.parametrize("shape,deg", [("aniso", 3), ("aniso", 1), ("iso", 3), ("clustered", 2), ("iso", 0)])
def test_sh_rows_read_where_the_level_lies_change_nothing(monkeypatch, shape, deg):
    """The M-step and the orphans' copy read a child's SH row from the level's OWN array through the child's input index (it rides in the
    record's flags word) instead of from a cell-sorted, padded copy of the whole SH block (k_gather_sh, rounds 1-4: GSR_HEM_SH_DIRECT=0).
    The rows of F = 45 / 9 floats are not 16-byte aligned there and the last float4 slot of a row runs into the next row (sums that are
    never stored) -- or, for the array's last row, would run behind the array: that one row is read from a padded copy.  Same products
    in the same order: three levels bit for bit, on clouds with many orphans (a third of a surfel level), heavy parents (segments) and
    small parents (four per wave), own buffers and zero-copy output (the level's array is then the caller's)."""
    from gaussiansplattingregistration_amd import hem, synth
    n = 300000 if shape == "aniso" else 150000
    c = synth.make_cloud(n, seed=74 if shape == "aniso" else 81, sh_degree=deg, shape=shape)
    if shape == "aniso":        # a few fat, nearly isotropic splats among the discs: parents of > 2048 pairs (segments); the array's last row among them
        g = np.append(np.random.default_rng(5).choice(n, 40, replace=False), n - 1)
        c["cov6"][g] = np.array([0.09, 0, 0, 0.08, 0, 0.07], np.float32)
    monkeypatch.setenv("GSR_HEM_SH_DIRECT", "0")
    with hem.HemMixture() as m:
        ref, rst = _levels_on_one_context(m, c, 3, False)
    monkeypatch.setenv("GSR_HEM_SH_DIRECT", "1")
    for zero_copy in (False, True):
        with hem.HemMixture() as m:
            got, st = _levels_on_one_context(m, c, 3, zero_copy)
        for k in range(3):
            assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"], rst[k]["dropped"])
            for f in ("xyz", "color", "cov6", "sh", "opacity", "weight"):
                assert np.array_equal(got[k][f].view(np.uint32), ref[k][f].view(np.uint32)), (zero_copy, k, f)
    if shape == "aniso" and deg == 3:
        assert rst[0]["orphans"] * 8 > rst[0]["n_in"] and max(s_["max_pairs_of_a_parent"] for s_ in rst) > 2048, [s_["max_pairs_of_a_parent"] for s_ in rst]


@pytest.mark.parametrize("deg", [3, 2, 1, 0])
def test_small_parent_path_changes_nothing(monkeypatch, deg):
    """Parents with at most 16 pairs are served four at a time, one per DPP row of the M-step's wavefront (most parents of a
    surfel-shaped cloud; GSR_HEM_MSTEP_SMALL=0: every parent takes the general path).  The sums are formed in the same order, so two
    levels are the same BIT FOR BIT -- on the surfel cloud (6 pairs per parent, thousands of orphans), on a sparse isotropic cloud
    whose parents have 0 .. 16 pairs, and on the dense one (hardly a small parent: the mixed case inside a wave), for every
    lanes-per-SH-row variant (SH degree 3, 2, 1 and no SH at all)."""
    from gaussiansplattingregistration_amd import hem, synth
    clouds = (synth.make_cloud(150000, seed=71, sh_degree=deg, shape="aniso"), synth.make_cloud(60000, seed=72, sh_degree=deg, h=2.6 * synth.half_extent(60000)),
              synth.make_cloud(80000, seed=73, sh_degree=deg))
    for ci, c in enumerate(clouds):
        monkeypatch.setenv("GSR_HEM_MSTEP_SMALL", "0")
        ref, rst = hem.create_mixture(c, 2, with_state=True)
        monkeypatch.setenv("GSR_HEM_MSTEP_SMALL", "1")
        got, st = hem.create_mixture(c, 2, with_state=True)
        for k in range(2):
            assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"], rst[k]["dropped"])
            for f in ("xyz", "color", "cov6", "sh", "opacity", "weight"):
                assert np.array_equal(got[k][f].view(np.uint32), ref[k][f].view(np.uint32)), (ci, k, f)
        if ci < 2:
            assert rst[0]["pairs"] < 17 * rst[0]["parents"]            # these clouds do have small parents


def test_mstep_heavy_parent_split_changes_nothing(monkeypatch):
    """A parent's sums are defined over segments of 2 048 pairs added in order (mstep_segment); a parent with more than one segment
    is cut into work items -- one wave per segment, a finish kernel adding them in order -- instead of keeping one wave busy for the
    whole M-step (a surfel cloud's largest parent has 5 * 10^4 pairs).  GSR_HEM_MSTEP_SPLIT=0 runs the segments in the parent's own
    wave: bit for bit the same levels.  The cloud is made to have such parents (checked)."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(300000, seed=74, sh_degree=3, shape="aniso")
    # a few fat, nearly isotropic splats of ordinary size variance among the discs: parents that many children merge into
    rng = np.random.default_rng(5)
    g = rng.choice(300000, 40, replace=False)
    c["cov6"][g] = np.array([0.09, 0, 0, 0.08, 0, 0.07], np.float32)
    res = {}
    for split in ("0", "1"):
        monkeypatch.setenv("GSR_HEM_MSTEP_SPLIT", split)
        res[split] = hem.create_mixture(c, 2, with_state=True)
    (ref, rst), (got, st) = res["0"], res["1"]
    assert max(s_["max_pairs_of_a_parent"] for s_ in st) > 2 * 2048, [s_["max_pairs_of_a_parent"] for s_ in st]
    for k in range(2):
        assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"]) == (rst[k]["parents"], rst[k]["pairs"], rst[k]["orphans"], rst[k]["dropped"])
        for f in ("xyz", "color", "cov6", "sh", "opacity", "weight"):
            assert np.array_equal(got[k][f].view(np.uint32), ref[k][f].view(np.uint32)), (k, f)


@pytest.mark.parametrize("shape,deg", [("iso", 3), ("aniso", 1), ("iso", 0)])
def test_zero_copy_level_output_equals_the_copied_levels(shape, deg):
    """``run_level(out=...)`` / ``gsr_hem_set_output``: a level is written straight into the caller's tensors, which then ARE the current
    level (borrowed) -- no copy out, none into the next level.  Three levels that way are bit for bit the levels of the copying path
    (the surfel cloud drops components in the validity erase: the compacted rows must land in the caller's arrays too); the returned
    tensors are views of the caller's, a too small output is refused cleanly, and the context carries on afterwards."""
    import torch
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(90000, seed=91, sh_degree=deg, shape=shape)
    if shape == "aniso":
        c["cov6"][11] = [1.0, 0, 0, 1.0, 0, -1.0]            # a component the validity erase drops
    want, wst = hem.create_mixture(c, 3, with_state=True)
    dc = {k: torch.from_numpy(c[k]).cuda() for k in ("xyz", "color", "opacity", "cov6", "sh")}
    with hem.HemMixture() as m:
        m.set_level0(dc["xyz"], dc["color"], dc["opacity"], dc["cov6"], dc["sh"], borrow=True)
        outs = []
        for k in range(3):
            out = m.new_output()
            n_out, dropped = m.run_level(out=out)
            got = m.get_level(as_torch=True, with_state=True)
            assert got["xyz"].data_ptr() == out["xyz"].data_ptr() and got["sh"].data_ptr() == out["sh"].data_ptr()      # views, not copies
            assert n_out == want[k]["xyz"].shape[0] and dropped == wst[k]["dropped"]
            outs.append(got)
        for k in range(3):          # every level is still intact in its own tensors after the later levels ran
            for f in ("xyz", "color", "cov6", "opacity", "sh", "weight", "is_parent"):
                assert np.array_equal(outs[k][f].cpu().numpy(), want[k][f]), (k, f)
        if shape == "aniso":
            assert sum(s_["dropped"] for s_ in wst) >= 1
        # an output that cannot hold the level: an error, nothing written past its end, and the context still works
        m.set_level0(dc["xyz"], dc["color"], dc["opacity"], dc["cov6"], dc["sh"])
        with pytest.raises(RuntimeError, match="output arrays hold"):
            m.run_level(out=m.new_output(rows=10))
        m.set_rng("glibc", 1, 0)                             # (the stream position of a fresh context, as `want` had it)
        m.set_level0(dc["xyz"], dc["color"], dc["opacity"], dc["cov6"], dc["sh"])
        m.run_level()
        assert np.array_equal(m.get_level()["xyz"], want[0]["xyz"])


def test_two_contexts_on_two_threads_equal_each_alone():
    """bench.py runs the HEM levels of the pair's two clouds side by side: two contexts on two streams, driven by two host threads
    (the C ABI releases the GIL; a context owns every buffer it touches, the error message is thread-local).  Each cloud's levels
    are bit for bit what its context computes alone."""
    import threading
    import torch
    from gaussiansplattingregistration_amd import hem, synth
    clouds = [synth.make_cloud(120000, seed=81, sh_degree=3), synth.make_cloud(90000, seed=82, sh_degree=1, shape="aniso")]
    alone = [hem.create_mixture(c, 3, with_state=True) for c in clouds]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctx = [hem.HemMixture(stream=s.cuda_stream) for s in streams]
    for rep in range(3):
        got, err = [None, None], []

        def work(k):
            try:
                with torch.cuda.stream(streams[k]):
                    m, c = ctx[k], clouds[k]
                    m.set_rng("glibc", 1, 0)
                    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
                    lv = []
                    for _ in range(3):
                        m.run_level()
                        lv.append(m.get_level(with_state=True))
                    got[k] = lv
            except BaseException as e:      # noqa
                err.append(e)

        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not err, err
        for k in range(2):
            for lvl in range(3):
                for f in ("xyz", "color", "cov6", "opacity", "sh", "weight", "is_parent"):
                    assert np.array_equal(got[k][lvl][f], alone[k][0][lvl][f]), (rep, k, lvl, f)
    for m in ctx:
        m.close()


def test_timing_levels_change_nothing_but_the_figures():
    """``gsr_hem_set_timing``: 0 records no event at all, 1 (the default) the level and the launches of k_select / k_mstep, 2 every
    phase.  Same level whatever the setting; the figures a setting does not record read 0; a value outside 0..2 is refused."""
    from gaussiansplattingregistration_amd import hem, synth, _lib
    c = synth.make_cloud(60000, seed=12, sh_degree=1)
    ref = None
    for level in (1, 0, 2):
        with hem.HemMixture() as m:
            if level != 1:
                m.set_timing(level)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            m.run_level()
            st = m.stats()
            got = m.get_level()
        if ref is None:
            ref = got
        for f in ("xyz", "color", "cov6", "sh", "opacity"):
            assert np.array_equal(got[f], ref[f]), (level, f)
        assert (st["ms_level"] > 0) == (level >= 1) and (st["ms_k_select"] > 0) == (level >= 1) and (st["ms_k_mstep"] > 0) == (level >= 1), (level, st)
        assert (st["ms_grid"] > 0) == (level == 2) and (st["ms_sumlw"] > 0) == (level == 2) and (st["ms_k_partition"] > 0) == (level == 2), (level, st)
    with hem.HemMixture() as m:
        with pytest.raises(RuntimeError, match="outside 0..2"):
            m.set_timing(3)


def test_read_back_poll_changes_nothing(monkeypatch):
    """The host reads counts back by polling a sequence word the device writes into pinned memory; GSR_HEM_RB_POLL=0 waits with
    hipStreamSynchronize instead.  Same values either way."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(20000, seed=36, h=1.1, sh_degree=1)
    a, sa = hem.create_mixture(c, 2)
    monkeypatch.setenv("GSR_HEM_RB_POLL", "0")
    b, sb = hem.create_mixture(c, 2)
    for k in range(2):
        assert (sa[k]["parents"], sa[k]["pairs"], sa[k]["orphans"]) == (sb[k]["parents"], sb[k]["pairs"], sb[k]["orphans"])
        for f in ("xyz", "color", "cov6", "sh", "opacity"):
            assert np.array_equal(a[k][f], b[k][f]), (k, f)


def test_grid_cell_size_changes_nothing(monkeypatch):
    """Any conservative neighbour search is legal: with 2 or 40 components per grid cell instead of 8
    (GSR_HEM_CELL_TARGET) a level accepts exactly the same pairs -- only the candidates scanned differ."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(30000, seed=33, h=1.2, sh_degree=1)
    ref, rst = hem.create_mixture(c, 1)
    for target in ("2", "40"):
        monkeypatch.setenv("GSR_HEM_CELL_TARGET", target)
        got, st = hem.create_mixture(c, 1)
        assert (st[0]["parents"], st[0]["pairs"], st[0]["orphans"]) == (rst[0]["parents"], rst[0]["pairs"], rst[0]["orphans"])
        assert st[0]["cells"] != rst[0]["cells"]
        for f in ("xyz", "cov6", "sh", "opacity"):
            assert _rel(got[0][f], ref[0][f]) < 1e-5, (target, f)


def test_fast_log_margin(hip_lib, oracle):
    """k_select decides the KL gate with det_c * (1 / det_p) and the hardware v_log_f32 when the value is farther from the
    threshold than the error bound of that path, and with the IEEE division + glibc's logf otherwise.  Check on the
    device (i) the bound the margin assumes -- |fast log - logf(det_c / det_p)| < 5e-7 (1 + |logf|) -- and (ii) that the
    decision equals the reference's float32 expression 0.5f * (s2 - logf(det_c / det_p)) > thr on random and on
    adversarial inputs (values placed within a few ulp of the threshold), and (iii) that the fallback is rare on ordinary
    inputs."""
    rng = np.random.default_rng(11)
    n, m = 400_000, 50_000
    det_p = np.exp(rng.normal(-14, 4, n)).astype(np.float32)
    det_c = (det_p * np.concatenate([np.exp(rng.uniform(-40, 40, n // 2)), np.exp(rng.normal(0, 2, n // 2))])).astype(np.float32)
    with np.errstate(all="ignore"):
        q = (det_c / det_p).astype(np.float32)               # IEEE float32 division, as the reference does it
        lq = np.array([oracle.logf(float(v)) for v in q[:m]], np.float32)
    lq_all = np.log(q.astype(np.float64)).astype(np.float32)
    lq_all[:m] = lq                                          # libm on the first 50 k, correctly rounded log elsewhere
    thr = np.float32(4.5)
    s2 = (2 * thr + lq_all + rng.normal(0, 3, n)).astype(np.float32)
    # adversarial part: k = 0.5 (s2 - log q) within a few ulp of thr
    s2[:m] = (np.float32(2) * thr + lq).astype(np.float32)
    s2[:m] = np.nextafter(s2[:m], (np.inf * rng.choice([-1.0, 1.0], m)).astype(np.float32)).astype(np.float32)
    special_c = np.float32([0.0, -1.0, np.inf, np.nan, 1e-45, 3e-39, 1.0, 1.0])
    special_s = np.float32([1.0, 1.0, 1.0, 1.0, 9.0, 9.0, np.inf, np.nan])
    det_c = np.concatenate([det_c, special_c]); det_p = np.concatenate([det_p, np.ones(8, np.float32)]); s2 = np.concatenate([s2, special_s])
    N = s2.size
    rej = np.empty(N, np.uint8); lf = np.empty(N, np.float32); ex = np.empty(N, np.uint8)
    assert hip_lib.gsr_debug_kl_gate(s2.ctypes.data, det_c.ctypes.data, det_p.ctypes.data, N, float(thr), rej.ctypes.data,
                                     lf.ctypes.data, ex.ctypes.data, 0) == 0
    # (i) error bound of the fast path, against libm where it was evaluated
    ok = np.isfinite(lq) & (q[:m] >= 4 * np.finfo(np.float32).tiny)
    err = np.abs(lf[:m].astype(np.float64) - lq.astype(np.float64))
    assert (err[ok] < 5e-7 * (1 + np.abs(lq[ok]))).all(), float((err[ok] / (1 + np.abs(lq[ok]))).max())
    # (ii) decisions: the reference's float32 expression with libm's logf
    with np.errstate(all="ignore"):
        k = np.float32(0.5) * (s2[:m] - lq)
    assert np.array_equal(rej[:m], (k > thr).astype(np.uint8))
    assert ex[:m].mean() > 0.9                               # the adversarial values do take the exact path
    with np.errstate(all="ignore"):
        k2 = np.float32(0.5) * (s2[m:n] - lq_all[m:n])
    far = np.abs(k2 - thr) > 1e-4                            # away from the threshold any correct logf decides alike
    assert np.array_equal(rej[m:n][far], (k2 > thr).astype(np.uint8)[far])
    # (iii) ordinary inputs: the fallback is rare
    assert ex[m:n].mean() < 1e-3
    # special values: log(0) = -inf -> k = +inf -> reject; q < 0 or NaN -> NaN -> passes; s2 = inf -> reject; s2 NaN -> passes
    assert list(rej[n:]) == [1, 0, 0, 0, 1, 1, 1, 0] and ex[n:].all()

def test_stage1_filter_never_rejects_an_accepted_pair(hip_lib, oracle):
    """The device's own stage-1 filter (gsr_debug_stage1: is_regular + make_filter + the nine fused multiply-adds of
    white_smd, the functions k_select runs) on 600 000 adversarial (parent, child) pairs just beyond each parent's bound:
    whatever it drops, the reference arithmetic rejects.  Its records agree with the NumPy restatement the CPU suite
    attacks (tests/stage1_model.py), and flat discs are inside the filter's reach (the round-2 predicate stopped at a
    condition number of 80)."""
    import stage1_model as S1
    from test_oracle_math import _adversarial_pairs
    thr = 4.5
    n = 600000
    pm, pc, cm, cc, F = _adversarial_pairs(n, 23, thr)
    preg = np.empty(n, np.uint8); creg = np.empty(n, np.uint8); white = np.empty(n, np.uint8); rej = np.empty(n, np.uint8)
    T1 = np.empty(n, np.float32); clip = np.empty(n, np.uint8)
    pm, pc, cm, cc = (np.ascontiguousarray(a, np.float32) for a in (pm, pc, cm, cc))
    assert hip_lib.gsr_debug_stage1(pm.ctypes.data, pc.ctypes.data, cm.ctypes.data, cc.ctypes.data, n, float(thr), preg.ctypes.data,
                                    creg.ctypes.data, white.ctypes.data, rej.ctypes.data, T1.ctypes.data, clip.ctypes.data, 0) == 0
    with np.errstate(all="ignore"):
        k = oracle.kld(cm, cc, pm, pc)
    bad = (rej == 1) & ~(k > thr)
    assert rej.sum() > 0.2 * n and not bad.any(), (int(rej.sum()), int(bad.sum()), np.flatnonzero(bad)[:5])
    # the device's records against the model: the predicates equal but for borderline float64 roundings, T1 to 1e-5
    mreg, _ = S1.is_regular(pc, pm)
    assert (preg.astype(bool) != mreg).mean() < 1e-4 and (white.astype(bool) != F["white"]).mean() < 1e-3
    both = white.astype(bool) & F["white"]
    assert np.abs(T1[both].astype(np.float64) / F["T1"][both] - 1).max() < 1e-5
    # reach: discs with condition numbers of 1e3 .. 1e4 are (mostly) certified and their rows clipped
    ev = np.linalg.eigvalsh(np.stack([pc[:, [0, 1, 2]], pc[:, [1, 3, 4]], pc[:, [2, 4, 5]]], 1).astype(np.float64))
    kap = ev[:, 2] / np.maximum(ev[:, 0], 1e-300)
    band = (kap > 1e3) & (kap < 1e4)
    assert white[band].mean() > 0.8 and clip[band].mean() > 0.8, (white[band].mean(), clip[band].mean())
    # not regular: never white, never rejected in stage 1
    assert not (white.astype(bool) & ~preg.astype(bool)).any() and not (rej.astype(bool) & ~creg.astype(bool)).any()


def test_anisotropic_cloud_equals_oracle(oracle):
    """The surfel workload (synth.make_cloud(shape="aniso"): 60 % flat discs, 15 % needles, condition numbers 1e2 .. 1e5 on a
    smooth orientation field): every discrete outcome of level 1 equals the oracle's -- parents, accepted pairs, orphans,
    dropped -- and the components to 1e-4; most components are inside the filter's precondition."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(150000, seed=4, shape="aniso")
    want, wst = oracle.hem(c, 2)
    with hem.HemMixture() as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        for k in range(2):
            m.run_level()
            st = m.stats()
            got = m.get_level()
            if k == 0:
                assert (st["parents"], st["pairs"], st["orphans"], st["dropped"]) == (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"])
                assert st["irregular"] < 0.25 * 150000, st["irregular"]
                _check_level(got, want[k], ("aniso", k))
            else:
                _check_level_mostly(got, want[k], ("aniso", k))


@pytest.mark.parametrize("deg,F", [(4, 72), (5, 105), (8, 240)])
def test_sh_wider_than_a_wavefront(oracle, deg, F):
    """F = 72 / 105 / 240 feature floats: the M-step instantiations with 8, 16 and 32 lanes per SH row -- the last one folds its sums the
    unpacked way (a row is wider than the 16 lanes a packed sum needs for itself).  A few fat splats make parents of several hundred
    pairs: more than one chunk of 128, so the per-chunk totals are added up on every path."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(3000, seed=41, h=0.6, sh_degree=deg)
    assert c["sh"].shape[1] == F
    c["cov6"][5:35:5] = np.array([0.05, 0, 0, 0.04, 0, 0.045], np.float32)
    want, wst = oracle.hem(c, 2)
    got, st = hem.create_mixture(c, 2)
    assert st[0]["max_pairs_of_a_parent"] > 128, st[0]
    for k in range(2):
        _check_level(got[k], want[k], ("F%d" % F, k))


def test_work_sharded_level_equals_single_gpu():
    """parallel.hem_sharded with 3 simulated ranks (threads sharing one GPU, an in-process all-reduce / all-gather standing
    in for RCCL) against the single-context result: equal counts, components within 1e-5.  (The multi-PROCESS version over
    torch.distributed is tests/test_distributed_gpu.py.)"""
    import threading
    from gaussiansplattingregistration_amd import hem, parallel, synth
    c = synth.make_cloud(40000, seed=51, sh_degree=1)
    want, wst = hem.create_mixture(c, 2)
    W = 3

    class FakeAllReduce:
        def __init__(self):
            self.parts = [None] * W
            self.bar = threading.Barrier(W)

        def make(self, rank):
            def fn(t):
                self.parts[rank] = t.clone()
                self.bar.wait()
                total = self.parts[0].clone()
                for r in range(1, W):
                    total += self.parts[r]          # fixed rank order: deterministic
                self.bar.wait()
                t.copy_(total)
            return fn

        def make_gather(self, rank):
            def fn(send, recv):
                self.parts[rank] = send.clone()
                self.bar.wait()
                allp = torch.cat([self.parts[r] for r in range(W)])
                self.bar.wait()
                recv.copy_(allp)
            return fn

    far = FakeAllReduce()
    out, errs = [None] * W, []

    def run(rank):
        try:
            out[rank] = parallel.hem_sharded(c, 2, rank, W, device=0, allreduce=far.make(rank), allgather=far.make_gather(rank))
        except Exception as e:      # pragma: no cover
            errs.append(e)
            far.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(W):
        levels, st = out[r]
        # every rank evaluated only its slab of parents ...
        assert st[0]["pairs"] < wst[0]["pairs"] and st[0]["parents"] == wst[0]["parents"]
        for k in range(2):
            # ... and holds the complete, identical level afterwards
            assert levels[k]["xyz"].shape == want[k]["xyz"].shape
            for f in ("xyz", "color", "cov6", "opacity", "sh"):
                assert _rel(levels[k][f], want[k][f]) < 1e-5, (r, k, f)
                assert np.array_equal(levels[k][f], out[0][0][k][f])
    assert sum(out[r][1][0]["pairs"] for r in range(W)) == wst[0]["pairs"]


def _needle_cloud(n, seed, frac=0.35, sh_degree=1):
    """A cloud where `frac` of the splats are irregular for the pre-reject (condition number >> 80: needles and
    discs, the shape real 3DGS splats often have), some with enormous anisotropy."""
    from gaussiansplattingregistration_amd import synth
    rng = np.random.default_rng(seed)
    c = synth.make_cloud(n, seed=seed, h=1.0, sh_degree=sh_degree)
    k = int(n * frac)
    idx = rng.choice(n, k, replace=False)
    s = np.exp(rng.normal(-2.5, 0.5, (k, 3)))
    s[:, 0] *= rng.choice([1e-2, 3e-2, 1e-3], k)              # squash one axis: kappa = 1e3 .. 1e6
    s[: k // 2, 1] *= 0.05                                      # half of them needles (two thin axes)
    q = rng.normal(size=(k, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    R = synth._quat_to_rot(q)
    L = R * s[:, None, :]
    C = (L @ L.transpose(0, 2, 1)).astype(np.float32)
    c["cov6"][idx] = C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
    return c


@pytest.mark.gpu
def test_many_irregular_components(oracle):
    """35 % of the components bypass the Mahalanobis pre-reject (second pass of k_select over the irregular list);
    regular parents clip their rows to the ellipsoid.  Discrete results equal the oracle's."""
    from gaussiansplattingregistration_amd import hem
    c = _needle_cloud(40000, seed=31)
    want, wst = oracle.hem(c, 2)
    with hem.HemMixture() as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        for k in range(2):
            m.run_level()
            st = m.stats()
            got = m.get_level()
            if k == 0:
                assert (st["parents"], st["pairs"], st["orphans"], st["dropped"]) == (wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"])
                _check_level(got, want[k], ("needles", k))
            else:
                _check_level_mostly(got, want[k], ("needles", k))


@pytest.mark.gpu
def test_ellipsoid_row_clipping_changes_nothing(monkeypatch):
    """Clipping a regular parent's grid rows to its pre-reject ellipsoid only skips candidates the pre-reject
    would discard: the pair count of a level is identical with the clipping switched off (GSR_HEM_ELL=0)."""
    from gaussiansplattingregistration_amd import hem, synth
    res = {}
    for ell in ("1", "0"):
        monkeypatch.setenv("GSR_HEM_ELL", ell)
        out = []
        for c in (synth.make_cloud(300000, seed=8), _needle_cloud(100000, seed=9, frac=0.2), synth.make_cloud(200000, seed=10, shape="aniso")):
            with hem.HemMixture() as m:
                m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
                m.run_level()      # one level: later levels inherit summation-order noise (the pair order differs)
                st = m.stats()
                out.append((st["parents"], st["pairs"], st["orphans"], st["dropped"], m.get_level()["xyz"].shape[0]))
        res[ell] = out
    assert res["1"] == res["0"], (res["1"], res["0"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_parameter_sweep_vs_oracle(oracle, seed):
    """Random (rho, delta, kappa, tau, SH degree, density, size) draws: every discrete outcome of level 1 equals the
    oracle's and the merged components agree; level 2 mostly (summation-order noise can flip a borderline pair)."""
    from gaussiansplattingregistration_amd import hem, synth
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3000, 25000))
    deg = int(rng.integers(0, 4))
    rho = float(rng.choice([1.5, 2.0, 3.0, 5.0, 10.0]))
    delta = float(rng.uniform(1.0, 4.0))
    kappa = float(rng.uniform(0.5, 4.0))
    tau = float(rng.uniform(0.3, 3.0))
    h = float(rng.uniform(0.6, 2.5))
    c = synth.make_cloud(n, seed=seed, h=h, sh_degree=deg)
    want, wst = oracle.hem(c, 2, rho=rho, delta=delta, kappa=kappa, tau=tau)
    with hem.HemMixture(hem_reduction=rho, distance_delta=delta, color_delta=kappa, decay_rate=tau) as m:
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
        for k in range(2):
            m.run_level()
            st = m.stats()
            got = m.get_level()
            tag = ("sweep", seed, k, n, deg, rho, round(delta, 2), round(kappa, 2), round(tau, 2))
            if k == 0:
                assert (st["parents"], st["pairs"], st["orphans"], st["dropped"]) == (
                    wst[0]["parents"], wst[0]["pairs"], wst[0]["orphans"], wst[0]["dropped"]), tag
                _check_level(got, want[k], tag)
            else:
                _check_level_mostly(got, want[k], tag)


@pytest.mark.gpu
def test_borrowed_level0_equals_copied():
    """set_level0(borrow=True) reads the caller's device tensors in place: identical levels to the copying path, the
    caller's tensors untouched, and the context still usable afterwards (its own buffers come back after the level)."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud_torch(300000, seed=9)
    ref = {k: c[k].clone() for k in ("xyz", "color", "opacity", "cov6", "sh")}
    outs = []
    for borrow in (False, True, True):
        with hem.HemMixture() as m:
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=borrow)
            lv = []
            for _ in range(2):
                m.run_level()
                lv.append(m.get_level())
            # the same context takes another (copied) cloud afterwards
            m.set_rng("glibc", 1, 0)
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            m.run_level()
            lv.append(m.get_level())
        outs.append(lv)
    for k in ("xyz", "color", "opacity", "cov6", "sh"):
        assert torch.equal(c[k], ref[k])
        for lvl in range(3):
            assert np.array_equal(outs[0][lvl][k], outs[1][lvl][k]) and np.array_equal(outs[1][lvl][k], outs[2][lvl][k])
    assert np.array_equal(outs[0][0]["xyz"], outs[0][2]["xyz"])          # the re-run after reset reproduces level 1


@pytest.mark.gpu
def test_bucketed_sums_equal_sorted_sums(monkeypatch):
    """The per-child sums of wL by bucket partition + LDS fixed point (default) against the radix-sort path
    (GSR_HEM_SUMLW=sort): same discrete outcome, components equal to float32 summation noise -- including children
    whose parents carry zero, tiny, huge, infinite and NaN weights -- and bit-identical from run to run."""
    from gaussiansplattingregistration_amd import hem, synth
    c = synth.make_cloud(120000, seed=12, h=1.6, sh_degree=1)
    rng = np.random.default_rng(3)
    w = np.ones(120000, np.float32)
    idx = rng.permutation(120000)
    w[idx[:200]] = 0.0
    w[idx[200:400]] = 1e-30
    w[idx[400:600]] = 1e30
    w[idx[600:620]] = np.inf
    w[idx[620:640]] = np.nan
    w[idx[640:660]] = -1.0                       # garbage input: negative weights
    res = {}
    for mode in ("bucket", "sort", "bucket"):
        monkeypatch.setenv("GSR_HEM_SUMLW", mode)
        with hem.HemMixture() as m:
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            m.set_state(weight=w)
            m.run_level()
            st = m.stats()
            res.setdefault(mode, []).append((st, m.get_level(with_state=True)))
    (sa, a), (sb, b), (sa2, a2) = res["bucket"][0], res["sort"][0], res["bucket"][1]
    for k in ("parents", "pairs", "orphans", "dropped", "n_out"):
        assert sa[k] == sb[k] == sa2[k], k
    for k in ("xyz", "color", "cov6", "opacity", "sh", "weight"):
        assert np.array_equal(a[k], a2[k], equal_nan=True), k                      # run-to-run: bit-identical
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y)), k
        ok = np.isfinite(x)
        scale = np.abs(y[ok]).max() + 1e-30
        assert np.abs(x[ok] - y[ok]).max() <= 2e-5 * scale, (k, np.abs(x[ok] - y[ok]).max() / scale)


@pytest.mark.gpu
def test_pair_partition_variants_change_nothing(monkeypatch):
    """The pairs go from the selection's per-parent segments straight into per-bucket regions of fixed capacity (k_partition);
    GSR_HEM_PARTITION=exact takes the histogram + scan + scatter of a compacted copy instead, and a region that overflows
    (GSR_HEM_PARTITION_FACTOR=0.3: capacity below the mean) makes the level redo its sums that way.  The per-child sums are order
    independent (LDS fixed point) and the M-step reads the same runs of pairs: three levels bit for bit identical, also on the
    two-pass fallback (compact CSR segments) and with many orphans (the anisotropic cloud).  GSR_HEM_PARTITION=walk: the
    partition kernel without its LDS staging (what chunks of heavy parents and levels beyond 23 M components use), GSR_HEM_PARTITION_STAGE:
    the smaller stages of the levels with more than 1 536 / 3 584 buckets;
    GSR_HEM_SH_DIRECT = 0 / 1: the cell-sorted copy of the SH block always / never made (by default the level's pair count decides)."""
    from gaussiansplattingregistration_amd import hem, synth
    clouds = [synth.make_cloud(250000, seed=21), synth.make_cloud(120000, seed=22, shape="aniso", sh_degree=1)]
    res = {}
    for tag, env in (("fixed", {}), ("exact", {"GSR_HEM_PARTITION": "exact"}), ("overflow", {"GSR_HEM_PARTITION_FACTOR": "0.3"}),
                     ("two-pass", {"GSR_HEM_SPARSE_GB": "0"}), ("walk", {"GSR_HEM_PARTITION": "walk"}), ("stage-6144", {"GSR_HEM_PARTITION_STAGE": "6144"}),
                     ("stage-4096", {"GSR_HEM_PARTITION_STAGE": "4096"}), ("sh-copy", {"GSR_HEM_SH_DIRECT": "0"}),
                     ("sh-direct", {"GSR_HEM_SH_DIRECT": "1"})):
        for k in ("GSR_HEM_PARTITION", "GSR_HEM_PARTITION_FACTOR", "GSR_HEM_PARTITION_STAGE", "GSR_HEM_SPARSE_GB", "GSR_HEM_SH_DIRECT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        out = []
        for c in clouds:
            with hem.HemMixture() as m:
                m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
                for _ in range(3):
                    m.run_level()
                    st = m.stats()
                    out.append((st["pairs"], st["orphans"], st["partition_overflow"], st["one_pass"], m.get_level(with_state=True)))
        res[tag] = out
    assert all(o[2] == 1 for o in res["overflow"][:3]) and not any(o[2] for o in res["fixed"])
    assert all(o[3] == 0 for o in res["two-pass"]) and all(o[3] == 1 for o in res["fixed"])
    for tag in ("exact", "overflow", "two-pass", "walk", "stage-6144", "stage-4096", "sh-copy", "sh-direct"):
        for a, b in zip(res["fixed"], res[tag]):
            assert a[:2] == b[:2], (tag, a[:2], b[:2])
            for f in ("xyz", "color", "cov6", "opacity", "sh", "weight", "is_parent"):
                assert np.array_equal(a[4][f], b[4][f]), (tag, f)


def test_survey_known_answers_on_the_gpu():
    """The reference's own numbers from SURVEY.md 8c, on the GPU: 100 k -> 33 142 components, 33 120 parents, 3 889 315
    pairs; 20 k (box +-5) -> 16 177 / 13 719; 200 k -> 66 405 / 22 070 / 7 435 (levels 2 and 3 may move by a pair or two:
    float32 summation order)."""
    from gaussiansplattingregistration_amd import hem, synth
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    lv, st = hem.create_mixture(synth.make_cloud(100000, seed=0, h=1.5), 1)
    s8 = ka["survey_8c"]["n100000_h1.5_seed0"]
    assert (lv[0]["xyz"].shape[0], st[0]["parents"], st[0]["pairs"]) == (s8["n_out"], s8["parents"], s8["pairs"])
    lv, st = hem.create_mixture(synth.make_cloud(20000, seed=0, h=5.0), 2)
    want = ka["counts_only"]["n20000_h5.0_seed0"]
    assert lv[0]["xyz"].shape[0] == want[0] and abs(lv[1]["xyz"].shape[0] - want[1]) <= 2
    lv, st = hem.create_mixture(synth.make_cloud(200000, seed=0, h=1.5), 3)
    want = ka["counts_only"]["n200000_h1.5_seed0"]
    assert lv[0]["xyz"].shape[0] == want[0]
    assert abs(lv[1]["xyz"].shape[0] - want[1]) <= 3 and abs(lv[2]["xyz"].shape[0] - want[2]) <= 3


def test_partitioned_level_with_one_rank_equals_the_plain_level():
    """The code path of the spatially partitioned level (grid from all-reduced box / histograms, sort in global-index order, split
    per-child sums, global ranks from bit maps, flags at global ranks) with a ONE-rank RCCL communicator -- every collective goes
    through librccl on the box's GPU: bit for bit the plain level, three levels, also with erased components."""
    from gaussiansplattingregistration_amd import hem, parallel, synth
    from gaussiansplattingregistration_amd.comm import Comm
    c = synth.make_cloud(80000, seed=71, sh_degree=2)
    c["cov6"][5] = [1.0, 0, 0, 1.0, 0, -1.0]
    want, wst = hem.create_mixture(c, 3)
    with Comm.rccl(Comm.unique_id(), 0, 1, 0) as cm:
        pieces, st = parallel.hem_partitioned(c, 3, cm, device=0)
    for k in range(3):
        assert np.array_equal(pieces[k]["gid"], np.arange(want[k]["xyz"].shape[0]))
        for f in ("xyz", "color", "cov6", "opacity", "sh"):
            assert np.array_equal(pieces[k][f], want[k][f]), (k, f)
        assert (st[k]["parents"], st[k]["pairs"], st[k]["orphans"], st[k]["dropped"]) == (wst[k]["parents"], wst[k]["pairs"], wst[k]["orphans"], wst[k]["dropped"])
        assert st[k]["ghosts"] == 0 and st[k]["n_global"] == want[k]["xyz"].shape[0]
