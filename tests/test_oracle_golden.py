"""The CPU oracle against the reference's golden vectors (tests/golden/*.npz were produced by the
reference's own compiled extension, see tests/golden/make_golden.py) -- bit for bit."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, HEM_CASES, golden_cloud, load_golden


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("case", HEM_CASES)
def test_oracle_bit_exact_vs_reference(oracle, case):
    g = load_golden(case)
    L = int(g["levels"])
    levels, stats = oracle.hem(golden_cloud(g), L, rho=float(g["rho"]), delta=float(g["delta"]), kappa=float(g["kappa"]),
                               tau=float(g["tau"]), rng_skip=int(g["pre_draws"]))
    for k in range(L):
        for f in ("xyz", "color", "opacity", "cov6", "sh"):
            want = g[f"out_{f}_{k}"]
            got = levels[k][f]
            assert got.shape == want.shape, (case, k, f, got.shape, want.shape)
            assert np.array_equal(_bits(got), _bits(want)), (case, k, f)


@pytest.mark.parametrize("case", HEM_CASES)
def test_oracle_fast_search_bit_exact_vs_reference(oracle, case):
    """The oracle's accelerated neighbour search (what tests/golden/make_golden_5m.py computes the 5 M digest with: the reference's result
    lists -- same members, same order -- found through a finer grid) against the reference's golden vectors, bit for bit, and equal to
    the plain 27-cell scan in every counter; clouds with non-finite coordinates (hem_edge) keep the plain scan."""
    g = load_golden(case)
    L = int(g["levels"])
    kw = dict(rho=float(g["rho"]), delta=float(g["delta"]), kappa=float(g["kappa"]), tau=float(g["tau"]), rng_skip=int(g["pre_draws"]))
    levels, stats = oracle.hem(golden_cloud(g), L, fast_search=True, **kw)
    _, plain = oracle.hem(golden_cloud(g), L, **kw)
    finite = bool(np.isfinite(g["xyz"]).all())
    for k in range(L):
        for f in ("xyz", "color", "opacity", "cov6", "sh"):
            assert np.array_equal(_bits(levels[k][f]), _bits(g[f"out_{f}_{k}"])), (case, k, f)
        for f in ("parents", "pairs", "orphans", "dropped", "candidates", "draws", "kld_margin", "color_margin"):
            assert stats[k][f] == plain[k][f], (case, k, f)
    if finite and stats[0]["parents"] > 0 and case != "hem_noparent":
        assert stats[0]["fast_search"], case


def test_oracle_fast_search_equals_plain_scan_on_every_cloud_shape(oracle):
    from gaussiansplattingregistration_amd import synth
    for shape, n in (("iso", 30000), ("aniso", 30000), ("clustered", 30000)):
        c = synth.make_cloud(n, seed=9, sh_degree=1, shape=shape)
        a, sa = oracle.hem(c, 2, fast_search=True)
        b, sb = oracle.hem(c, 2)
        assert sa[0]["fast_search"] and not sb[0]["fast_search"]
        for k in range(2):
            assert sa[k]["candidates"] == sb[k]["candidates"] and sa[k]["pairs"] == sb[k]["pairs"]
            for f in a[k]:
                assert a[k][f].tobytes() == b[k][f].tobytes(), (shape, k, f)


def test_known_answer_counts_and_parent_mask(oracle):
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    assert ka["hem_deg3"] == [537, 185, 59]
    # parent-mask KAT: 1000 splats, rho=3, fresh process -> 373 parents; with delta=0.01 nothing
    # merges, so the level is parents ++ non-parents in input order
    g = load_golden("hem_tiny_delta")
    mask = oracle.parent_flags(1000, 3.0).astype(bool)
    assert int(mask.sum()) == 373
    want = np.concatenate([g["xyz"][mask], g["xyz"][~mask]])
    assert np.array_equal(g["out_xyz_0"], want)


def test_edge_case_behaviour_recorded_by_reference():
    # rho=1: count unchanged; rho=1e9 (no parents): level returned unchanged, bit for bit
    g = load_golden("hem_rho1")
    assert g["out_xyz_0"].shape[0] == 500 and g["out_xyz_1"].shape[0] == 500
    g = load_golden("hem_noparent")
    for f in ("xyz", "color", "opacity", "cov6", "sh"):
        assert np.array_equal(g[f"out_{f}_0"], g[f]), f
    # degenerate inputs are dropped by the validity erase, never propagated
    g = load_golden("hem_edge")
    assert np.isfinite(g["out_xyz_0"]).all() and np.isfinite(g["out_cov6_0"]).all()


def test_oracle_stats_and_conservation(oracle):
    """Sum of weights is conserved by a level up to dropped components (responsibilities sum to 1)."""
    from gaussiansplattingregistration_amd import synth
    c = synth.make_cloud(4000, seed=2, h=0.65, sh_degree=1)
    o = oracle.HemOracle(c["xyz"], c["color"], c["cov6"], c["opacity"], c["sh"])
    n1 = o.run_level()
    st = o.stats()
    lv0, lv1 = o.level(0), o.level(1)
    assert n1 == st["parents"] + st["orphans"] - st["dropped"]
    assert st["dropped"] == 0
    assert abs(float(lv1["weight"].sum(dtype=np.float64)) - float(lv0["weight"].sum(dtype=np.float64))) < 1e-3 * 4000
    # orphans are copied unchanged, after all parents, in input order
    P = st["parents"]
    orph = lv1["xyz"][P:]
    assert all(any(np.array_equal(x, y) for y in lv0["xyz"]) for x in orph[:5])
    o.close()


def test_survey_known_answers_of_the_reference(oracle):
    """Numbers the survey measured on the reference's own extension: 100 k splats in the +-1.5 box ->
    33 142 components (33 120 parents, 3 889 315 accepted pairs); 20 k splats in the +-5 box -> 16 177 / 13 719."""
    from gaussiansplattingregistration_amd import synth
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
    w, st = oracle.hem(synth.make_cloud(100000, seed=0, h=1.5), 1)
    s8 = ka["survey_8c"]["n100000_h1.5_seed0"]
    assert (w[0]["xyz"].shape[0], st[0]["parents"], st[0]["pairs"]) == (s8["n_out"], s8["parents"], s8["pairs"])
    w, _ = oracle.hem(synth.make_cloud(20000, seed=0, h=5.0), 2)
    assert [x["xyz"].shape[0] for x in w] == ka["counts_only"]["n20000_h5.0_seed0"]
