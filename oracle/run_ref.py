#!/usr/bin/env python3
"""Run the REFERENCE's own compiled HEM extension (oracle/_ref/mixture_bind*.so) on one input.

TEST INFRASTRUCTURE ONLY.  The .so is built by ``make -C oracle ref`` from the reference's six
C++ sources where they lie under /root/reference (``setup.py:7-10``); nothing of the reference is
stored in this repo.  This script is the call pattern of the reference's own worker
(``src/gui/workers/downsampling/qt_gaussian_mixture.py:42-58,96``):

    level0 = MixtureLevel.CreateMixtureLevel(xyz, colors, opacities, cov6, features)   # python lists
    levels = MixtureCreator.CreateMixture(L, rho, delta, kappa, tau, level0)           # level 0 dropped
    xyz, colors, opacities, cov6, features = MixtureLevel.CreatePythonLists(levels[k])

It must be started as a FRESH PROCESS per case: parent sampling uses the process-global libc
``rand()`` (``src/cpp_ext/include/base.hpp:44-56``) which is never seeded, so only a fresh process
replays the seed-1 stream.  ``--pre-draws K`` burns K ``hem::rand()`` values first by running a
K-splat dummy mixture, which is how the "second cloud continues the first cloud's stream"
behaviour (``qt_gaussian_mixture.py:55,79``) is exercised.

usage: run_ref.py in.npz out.npz [--threads T]
  in.npz : xyz (n,3) color (n,3) opacity (n,) cov6 (n,6) sh (n,F) f32; levels, rho, delta, kappa, tau
  out.npz: n_levels, and per level k: xyz_k color_k opacity_k cov6_k sh_k ; wall_s (CreateMixture only)
"""
import argparse
import gc
import glob
import importlib.util
import os
import sys
import time

import numpy as np


def load_ref():
    here = os.path.dirname(os.path.abspath(__file__))
    cands = glob.glob(os.path.join(here, "_ref", "mixture_bind*.so"))
    if not cands:
        raise RuntimeError("oracle/_ref/mixture_bind*.so missing: run `make -C oracle ref` in the build container")
    spec = importlib.util.spec_from_file_location("mixture_bind", cands[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(mb, d, levels, rho, delta, kappa, tau):
    gc.disable()                      # list marshalling is super-linear with the cyclic GC on
    xyz = d["xyz"].astype(np.float32).tolist()
    col = d["color"].astype(np.float32).tolist()
    op = d["opacity"].astype(np.float32).reshape(-1).tolist()
    cov = d["cov6"].astype(np.float32).tolist()
    sh = d["sh"].astype(np.float32).tolist()
    t0 = time.perf_counter()
    ml = mb.MixtureLevel.CreateMixtureLevel(xyz, col, op, cov, sh)
    t1 = time.perf_counter()
    res = mb.MixtureCreator.CreateMixture(int(levels), float(rho), float(delta), float(kappa), float(tau), ml)
    t2 = time.perf_counter()
    out = {"n_levels": np.int64(len(res)), "wall_s": np.float64(t2 - t1), "marshal_s": np.float64(t1 - t0)}
    F = d["sh"].shape[1]
    for k, lv in enumerate(res):
        x, c, o, cv, f = mb.MixtureLevel.CreatePythonLists(lv)
        n = len(x)
        out[f"xyz_{k}"] = np.asarray(x, dtype=np.float32).reshape(n, 3)
        out[f"color_{k}"] = np.asarray(c, dtype=np.float32).reshape(n, 3)
        out[f"opacity_{k}"] = np.asarray(o, dtype=np.float32).reshape(n)
        out[f"cov6_{k}"] = np.asarray(cv, dtype=np.float32).reshape(n, 6)
        out[f"sh_{k}"] = np.asarray(f, dtype=np.float32).reshape(n, F)
    gc.enable()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("inp")
    ap.add_argument("out")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--pre-draws", type=int, default=0)
    a = ap.parse_args()
    if a.threads > 0:
        os.environ["OMP_NUM_THREADS"] = str(a.threads)
    mb = load_ref()
    d = dict(np.load(a.inp))
    if a.pre_draws > 0:
        # a K-splat throw-away mixture with zero levels draws exactly K hem::rand() values (initMixture)
        k = a.pre_draws
        z3 = np.zeros((k, 3), np.float32).tolist()
        ml = mb.MixtureLevel.CreateMixtureLevel(z3, z3, [0.0] * k, np.tile(np.float32([1, 0, 0, 1, 0, 1]), (k, 1)).tolist(),
                                                np.zeros((k, 1), np.float32).tolist())
        mb.MixtureCreator.CreateMixture(0, 3.0, 3.0, 2.5, 1.0, ml)
    # the reference prints "clustering level k" on stdout and dropped components on stderr
    out = run(mb, d, d["levels"], d["rho"], d["delta"], d["kappa"], d["tau"])
    np.savez(a.out, **out)
    print("levels:", [int(out[f"xyz_{k}"].shape[0]) for k in range(int(out["n_levels"]))],
          "wall %.3fs marshal %.3fs" % (out["wall_s"], out["marshal_s"]), file=sys.stderr)


if __name__ == "__main__":
    main()
