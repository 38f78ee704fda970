"""Randomised sweep over TINY and PATHOLOGICAL clouds (test infrastructure, not collected by pytest): 1 .. 400 splats with duplicated
points, zero / negative-determinant / NaN / inf covariances, NaN and inf coordinates, huge and tiny scales, rho from 1 (all parents)
to 1e9 (none), F = 0 .. 45 -- two levels on the GPU against the oracle: level sizes and the discrete counters EQUAL, values equal
to 1e-4 where finite and NaN / inf in the same places.  usage: python tests/stress_tiny.py [cases] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gaussiansplattingregistration_amd import hem, synth
from oracle import oracle as O


def same(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return False
    if not np.array_equal(np.isnan(a), np.isnan(b)) or not np.array_equal(a[~fa & ~np.isnan(a)], b[~fb & ~np.isnan(b)]):
        return False
    if not fa.any():
        return True
    scale = np.abs(b[fb]).max() + 1e-30
    return bool(np.abs(a[fa] - b[fb]).max() / scale < 1e-4)


def sweep(cases=200, seed=5, log=print):
    """-> number of failing clouds"""
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for k in range(cases):
        n = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 130, 400]))
        deg = int(rng.choice([0, 0, 1, 3]))
        c = synth.make_cloud(n, seed=int(rng.integers(1 << 30)), h=float(rng.choice([0.05, 0.3, 1.0])), sh_degree=deg)
        what = []
        def pick(frac):
            m = max(1, int(n * frac))
            return rng.choice(n, min(n, m), replace=False)
        if rng.random() < 0.3: i = pick(0.2); c["xyz"][i] = c["xyz"][rng.choice(n, len(i))]; what.append("dup")
        if rng.random() < 0.2: c["cov6"][pick(0.1)] = 0.0; what.append("cov0")
        if rng.random() < 0.2: c["cov6"][pick(0.1)] = np.array([1, 0, 0, 1, 0, -1], np.float32) * 0.01; what.append("det<0")
        if rng.random() < 0.15: c["cov6"][pick(0.05), int(rng.integers(6))] = np.nan; what.append("covNaN")
        if rng.random() < 0.15: c["xyz"][pick(0.05), int(rng.integers(3))] = np.nan; what.append("xyzNaN")
        if rng.random() < 0.1: c["xyz"][pick(0.05), int(rng.integers(3))] = np.inf; what.append("xyzInf")
        if rng.random() < 0.15: c["cov6"][pick(0.1)] *= np.float32(1e8); what.append("huge")
        if rng.random() < 0.15: c["cov6"][pick(0.1)] *= np.float32(1e-12); what.append("tiny")
        if rng.random() < 0.1: c["opacity"][pick(0.1)] = np.nan; what.append("opNaN")
        if rng.random() < 0.1: c["color"][pick(0.1), 0] = np.inf; what.append("colInf")
        rho = float(rng.choice([1.0, 1.5, 3.0, 3.0, 10.0, 1e9]))
        p = dict(rho=rho, delta=float(rng.choice([1.0, 3.0, 6.0])), kappa=float(rng.choice([0.5, 2.5, 10.0])), tau=float(rng.choice([0.3, 1.0])))
        try:
            want, wst = O.hem(c, 2, **p)
            with hem.HemMixture(hem_reduction=p["rho"], distance_delta=p["delta"], color_delta=p["kappa"], decay_rate=p["tau"]) as m:
                m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
                got, gst = [], []
                for _ in range(2):
                    m.run_level(); gst.append(m.stats()); got.append(m.get_level())
            ok, why = True, ""
            for lvl in range(2):
                a = tuple(gst[lvl][q] for q in ("parents", "pairs", "orphans", "dropped"))
                b = tuple(wst[lvl][q] for q in ("parents", "pairs", "orphans", "dropped"))
                if a != b:
                    ok, why = False, f"level {lvl + 1} counters {a} vs {b}"; break
                for f in ("xyz", "color", "cov6", "opacity", "sh"):
                    if not same(got[lvl][f], want[lvl][f]):
                        ok, why = False, f"level {lvl + 1} {f}"; break
                if not ok:
                    break
        except Exception as e:      # noqa
            ok, why = False, f"exception {type(e).__name__}: {str(e)[:120]}"
        bad += 0 if ok else 1
        if not ok:
            log(f"FAIL {k} n={n} deg={deg} {what} {p}: {why}")
    log(f"{cases - bad} of {cases} tiny / pathological clouds equal the oracle ({time.time() - t0:.0f} s)")
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    sys.exit(1 if sweep(cases, seed, log=lambda s: print(s, flush=True)) else 0)


if __name__ == "__main__":
    main()
