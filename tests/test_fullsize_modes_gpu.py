"""BASELINE configs[3] / configs[4] with their full rank counts on the test box's ONE GPU (scripts/fullsize_modes.py): at 1 M splats per
rank AND at the workload's own size -- 40 M splats over 8 ranks, 2 x 5 M over 2 (round 5 ran these by hand only:
profiles/r05_c5_40m_8ranks.json, r05_c4_2x5m.json):

* c5: 8 processes, each holding only its block of an 8 M-splat cloud, three spatially partitioned HEM levels through the library's RCCL
  branch (tests/mock_rccl stands in for librccl: RCCL refuses several ranks on one device) -- the assembled levels must be BIT FOR BIT the
  levels ONE context computes from the concatenated blocks (a 64-bit hash per array and level, keyed by the rows' global indices);
* c4: 2 processes, cloud A's levels on rank 0 and cloud B's on rank 1, ICP with the source split + the library's all-reduce -- the final
  transform must equal the single-process registration's to 1e-9 with the same iteration counts on every level."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = os.path.join(ROOT, "scripts", "fullsize_modes.py")


def _run(args, tmp_path):
    out = str(tmp_path / "rec.json")
    r = subprocess.run([sys.executable, SCRIPT] + args + ["--out", out, "--timeout", "600"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.load(open(out))


def test_c5_eight_ranks_one_million_each_bit_identical(tmp_path):
    d = _run(["c5", "--splats", "8000000", "--world", "8"], tmp_path)
    assert d["ok"] and d["bit-identical"] is True and d["transport"] == "rccl" and d["world"] == 8
    assert len(d["compare"]) == 3 and all(c["bit_identical"] for c in d["compare"])
    assert d["compare"][0]["n_in_global"] == 8000000 and d["compare"][0]["rows_over_ranks"] == d["compare"][0]["n_out_global"]
    for r in d["ranks"]:
        l1 = r["levels"][0]
        assert l1["owned_in"] == 1000000 and 0 < l1["ghosts"] < 3 * l1["owned_in"]
        assert l1["halo_bytes_received"] == l1["ghosts"] * (72 + 4 * 45) and l1["sum_exchange_bytes_received"] > 0
        assert not l1["partition_overflow"] and l1["one_pass"]


def test_c4_two_ranks_one_million_equal_to_single_process(tmp_path):
    d = _run(["c4", "--splats", "1000000"], tmp_path)
    assert d["ok"] and d["transform_equal_1e-9"] and d["iterations_equal"] and d["transport"] == "rccl"
    assert d["single_process"]["level_sizes"][0] == 1000000 and d["single_process"]["T_err_vs_ground_truth_F"] < 1e-3
    assert max(d["T_diff_F_vs_single_process"]) <= 1e-9


def test_c5_full_size_40m_over_eight_ranks_bit_identical(tmp_path):
    """BASELINE configs[4] at its own size: a 40 M-splat cloud, 5 M per rank, three spatially partitioned levels = the levels ONE context
    computes from the concatenated blocks, bit for bit (every array of every level)."""
    d = _run(["c5", "--splats", "40000000", "--world", "8"], tmp_path)
    assert d["ok"] and d["bit-identical"] is True and d["transport"] == "rccl" and d["world"] == 8
    assert len(d["compare"]) == 3 and all(c["bit_identical"] for c in d["compare"])
    assert d["compare"][0]["n_in_global"] == 40000000 and d["compare"][0]["rows_over_ranks"] == d["compare"][0]["n_out_global"]
    sizes = [c["n_out_global"] for c in d["compare"]]
    assert all(0.30 * a < b < 0.37 * a for a, b in zip([40000000] + sizes, sizes)), sizes
    for r in d["ranks"]:
        l1 = r["levels"][0]
        assert l1["owned_in"] == 5000000 and 0.1 * l1["owned_in"] < l1["ghosts"] < 0.6 * l1["owned_in"]        # 2 x 2 x 2 blocks: ~27 % ghosts
        assert l1["halo_bytes_received"] == l1["ghosts"] * (72 + 4 * 45) and l1["sum_exchange_bytes_received"] > 0
        assert not l1["partition_overflow"] and l1["one_pass"]


def test_c4_full_size_2x5m_equal_to_single_process(tmp_path):
    """BASELINE configs[3] at its own size on bench.py's own pair: cloud A's levels on rank 0, cloud B's on rank 1 (each from its own rand()
    stream position, bench.C4_STREAM_STRIDE), ICP with the source split over both ranks and the library's all-reduce of 32 float64 per
    iteration -- final transform = the single process's to 1e-9 with equal iteration counts, and within 1e-3 of the ground truth."""
    d = _run(["c4", "--splats", "5000000"], tmp_path)
    assert d["ok"] and d["transform_equal_1e-9"] and d["iterations_equal"] and d["transport"] == "rccl"
    sp = d["single_process"]
    assert sp["level_sizes"][0] == 5000000 and sp["T_err_vs_ground_truth_F"] < 1e-3 and sp["fitness"] > 0.99
    assert sp["iterations"][0] == 50 and sum(sp["iterations"]) > 60, sp["iterations"]      # independent flags: the finer entries still have work to do
    assert max(d["T_diff_F_vs_single_process"]) <= 1e-9
