"""bench.py end to end at a small size: the default (replica) line and the multi-GPU modes of BASELINE configs[3] / [4]
(`--mode c4`, `--mode c5`), the latter with two ranks sharing the one GPU of the test box over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _line(out):
    rows = [l for l in out.splitlines() if l.startswith("{")]
    assert rows, out[-2000:]
    return json.loads(rows[-1])


def _torchrun(args, nproc=2, port=29711):
    env = dict(os.environ, GSR_BENCH_SAME_DEVICE="1", GSR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r.stdout)


def test_default_line_small():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--splats", "200000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["unit"] == "Gaussians/s" and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 1e6 and d["config"]["level_sizes"][0] == 200000
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and 0 < ro["frac"] < 1 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert ro["level1"]["n_in"] == 200000 and ro["icp_finest"]["ns"] == 200000
    assert d["icp_result"]["T_err_vs_ground_truth_F"] < 1e-3


def test_mode_c4_two_ranks():
    d = _torchrun(["--mode", "c4", "--splats", "150000", "--steps", "1", "--warmup", "1"], port=29713)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["mode"] == "c4"
    assert d["config"]["level_sizes"][0] == 150000 and d["icp_result"]["T_err_vs_ground_truth_F"] < 1e-3
    assert d["exchange_s_per_step"] > 0


def test_mode_c5_two_ranks():
    d = _torchrun(["--mode", "c5", "--splats", "300000", "--target-splats", "100000", "--steps", "1", "--warmup", "0"], port=29715)
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "c5" and d["config"]["level_sizes"][0] == 300000
    # the two ranks share the levels: counted once (3 levels of the 300 k cloud + 3 of the 100 k target)
    assert 1.3 * 400000 < d["value"] * d["hem_s_per_step"] < 1.6 * 400000
    assert np.isfinite(d["icp_result"]["T_err_vs_ground_truth_F"])
