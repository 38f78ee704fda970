"""A fixed-seed slice of every randomised sweep (tests/stress_*.py) under `-m gpu`, so that the driver's GPU tier sees them
(VERDICT r03 "what's weak" 3).  The full sweeps are run through gpurun and their logs are kept under profiles/archive/r04_stress_*.txt (round 5, after the M-step and orphan-copy changes: r05_stress_*.txt).

Outcome classes: `ok` = discrete outcomes EQUAL to the oracle and values within 1e-4; `edge` = exactly the one documented class --
parents, accepted pairs and orphans equal, the validity erase of a near-singular MERGED covariance (det <= 0 of a float32 sum,
mixture.cpp:262-274) off by at most 2 rows (8 with injected pathologies); anything else fails the test.
Reference semantics: src/cpp_ext/src/mixture.cpp:102-137,262-282; src/utils/local_registration_util.py:76-100."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _collect():
    lines = []
    return lines, lines.append


def test_parity_sweep_slice(oracle, hip_lib):
    import stress_parity
    lines, log = _collect()
    ok, edge, bad = stress_parity.sweep(20, 2026, log=log, pathologies=False)
    assert bad == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert edge <= 1, "\n".join(l for l in lines if l.startswith("edge"))         # 6 in 400 over the full sweep
    assert ok + edge == 20


def test_parity_sweep_slice_with_pathologies(oracle, hip_lib):
    import stress_parity
    lines, log = _collect()
    ok, edge, bad = stress_parity.sweep(10, 77, log=log, pathologies=True)
    assert bad == 0, "\n".join(l for l in lines if l.startswith("FAIL"))
    assert edge <= 1, "\n".join(l for l in lines if l.startswith("edge"))
    assert ok + edge == 10


def test_tiny_and_pathological_sweep_slice(oracle, hip_lib):
    import stress_tiny
    lines, log = _collect()
    assert stress_tiny.sweep(200, 5, log=log) == 0, "\n".join(lines[-8:])


def test_knob_sweep_slice(hip_lib):
    import stress_knobs
    lines, log = _collect()
    try:
        assert stress_knobs.sweep(4, 31, log=log) == 0, "\n".join(lines)
    finally:
        for k in stress_knobs.ALL:
            os.environ.pop(k, None)


def test_icp_sweep_slice(oracle, hip_lib):
    import stress_icp
    lines, log = _collect()
    bad = stress_icp.sweep(40, 7, log=log)
    assert bad == [], "\n".join(l for l in lines if l.startswith("FAIL"))


def test_partition_sweep_slice(hip_lib):
    import stress_partition
    lines, log = _collect()
    assert stress_partition.sweep(8, 3, 99, log=log) == 0, "\n".join(lines)
