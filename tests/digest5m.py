"""Digest of one HEM level of a LARGE cloud: what tests/golden/hem_5m_digest.npz holds of the reference's level 1 of the bench's own
5 M-splat cloud, and what the -m gpu test recomputes from the HIP level (tests/test_configs_gpu.py).  Shared by the generator
(tests/golden/make_golden_5m.py) and the test so that both sides are reduced by the same code.

A level of 1.67 M rows x 237 bytes does not fit a fixture; the digest keeps
  * the discrete outcome: rows, parents, accepted pairs, orphans, dropped rows, rand() draws, and the parent flags of EVERY row (bit-packed);
  * global moments in float64: sum of weights, weighted mean, weighted total covariance, weighted mean colour / opacity / SH, RMS scales;
  * SAMPLE_ROWS output rows chosen by a seeded generator, every array of them (xyz, colour, cov6, opacity, SH, weight);
  * per block of BLOCK consecutive output rows the float64 sums of every column (and of |column|, the scale its tolerance refers to): a
    single wrong row anywhere in the level moves its block's sums by far more than the tolerance, so the blocks cover what the sample skips.
"""
import hashlib

import numpy as np

SAMPLE_ROWS = 10_000
BLOCK = 1024
FIELDS = ("xyz", "color", "cov6", "opacity", "sh", "weight")


def input_hash(cloud):
    """sha256 over the five input arrays: the GPU box must have drawn the very cloud the fixture was computed from."""
    h = hashlib.sha256()
    for f in ("xyz", "color", "opacity", "cov6", "sh"):
        h.update(np.ascontiguousarray(cloud[f], dtype=np.float32).tobytes())
    return h.hexdigest()


def sample_index(n_out, seed=2024, rows=SAMPLE_ROWS):
    return np.sort(np.random.default_rng(seed).choice(n_out, size=min(rows, n_out), replace=False))


def _cols(lv):
    """(n, 3+3+6+1+F+1) float64 matrix of every exported column + the weight."""
    n = lv["xyz"].shape[0]
    return np.concatenate([np.asarray(lv["xyz"], np.float64).reshape(n, 3), np.asarray(lv["color"], np.float64).reshape(n, 3),
                           np.asarray(lv["cov6"], np.float64).reshape(n, 6), np.asarray(lv["opacity"], np.float64).reshape(n, 1),
                           np.asarray(lv["sh"], np.float64).reshape(n, -1), np.asarray(lv["weight"], np.float64).reshape(n, 1)], 1)


def global_moments(lv):
    w = np.asarray(lv["weight"], np.float64)
    W = w.sum()
    x = np.asarray(lv["xyz"], np.float64)
    c6 = np.asarray(lv["cov6"], np.float64)
    mean = (w[:, None] * x).sum(0) / W
    d = x - mean
    outer = np.stack([d[:, 0] * d[:, 0], d[:, 0] * d[:, 1], d[:, 0] * d[:, 2], d[:, 1] * d[:, 1], d[:, 1] * d[:, 2], d[:, 2] * d[:, 2]], 1)
    col = np.asarray(lv["color"], np.float64)
    op = np.asarray(lv["opacity"], np.float64)
    sh = np.asarray(lv["sh"], np.float64)
    return {"W": W, "mean": mean, "cov": (w[:, None] * (c6 + outer)).sum(0) / W, "color": (w[:, None] * col).sum(0) / W,
            "opacity": (w * op).sum() / W, "sh": (w[:, None] * sh).sum(0) / W,
            "rms_color": np.sqrt((w[:, None] * col * col).sum() / W / 3), "rms_opacity": np.sqrt((w * op * op).sum() / W),
            "rms_sh": np.sqrt((w[:, None] * sh * sh).sum() / W / max(1, sh.shape[1])), "extent": np.abs(x).max()}


def block_sums(lv):
    """Per block of BLOCK consecutive rows: column sums and sums of absolute values, float64."""
    M = _cols(lv)
    n, k = M.shape
    nb = (n + BLOCK - 1) // BLOCK
    pad = nb * BLOCK - n
    if pad:
        M = np.concatenate([M, np.zeros((pad, k))], 0)
    M = M.reshape(nb, BLOCK, k)
    return M.sum(1), np.abs(M).sum(1)


def digest(lv, stats, idx=None):
    """lv: dict of numpy arrays (xyz, color, cov6, opacity, sh, weight, is_parent); stats: parents / pairs / orphans / dropped / draws."""
    n = lv["xyz"].shape[0]
    idx = sample_index(n) if idx is None else idx
    g = global_moments(lv)
    bs, ba = block_sums(lv)
    out = {"n_out": np.int64(n), "sample_idx": idx.astype(np.int64), "flags_packed": np.packbits(np.asarray(lv["is_parent"], np.uint8)),
           "block_sum": bs, "block_abs": ba}
    for k in ("parents", "pairs", "orphans", "dropped", "draws"):
        out[k] = np.int64(stats[k])
    for k, v in g.items():
        out["g_" + k] = np.asarray(v, np.float64)
    for f in FIELDS:
        out["s_" + f] = np.asarray(lv[f])[idx]
    return out


def compare(got, want, tol=1e-4):
    """Differences between two digests (got: the HIP level's, want: the fixture's) as a list of failure strings (empty = equal).
    Counts and flags exact; global moments, block sums and sampled rows to `tol` -- every sampled row on ITS OWN scale:
      position     against the row's extent sqrt(trace of its covariance),
      covariance   all six entries against the row's trace,
      colour / SH  against the row's own largest |entry| of that array, floored at 5 % of the field's RMS (zero-mean fields: an entry
                   that cancels far below the field's scale is a float32 sum of terms tens to thousands of times its size, and is not
                   asked for more digits than float32 carried through that sum),
      opacity, weight   against their own magnitude (same floor)."""
    bad = []
    for k in ("n_out", "parents", "pairs", "orphans", "dropped", "draws"):
        if int(got[k]) != int(want[k]):
            bad.append(f"{k}: {int(got[k])} != {int(want[k])}")
    if bad:
        return bad
    if not np.array_equal(got["flags_packed"], want["flags_packed"]):
        bad.append("parent flags of the new level differ")
    W = float(want["g_W"])
    chk = [("W", abs(float(got["g_W"]) - W), W), ("mean", np.abs(got["g_mean"] - want["g_mean"]).max(), float(want["g_extent"])),
           ("cov", np.abs(got["g_cov"] - want["g_cov"]).max(), np.abs(want["g_cov"]).max()),
           ("color", np.abs(got["g_color"] - want["g_color"]).max(), float(want["g_rms_color"])),
           ("opacity", abs(float(got["g_opacity"]) - float(want["g_opacity"])), float(want["g_rms_opacity"])),
           ("sh", np.abs(got["g_sh"] - want["g_sh"]).max(), float(want["g_rms_sh"]))]
    for name, err, scale in chk:
        if not err <= tol * scale:
            bad.append(f"global moment {name}: |d| {err:.3e} > {tol} x {scale:.3e}")
    e = np.abs(got["block_sum"] - want["block_sum"]) / np.maximum(want["block_abs"], 1e-30)
    if not e.max() <= tol:
        b, c = np.unravel_index(int(np.argmax(e)), e.shape)
        bad.append(f"block sums: block {b} column {c} off by {e.max():.3e} of its sum of magnitudes")
    if not np.array_equal(got["sample_idx"], want["sample_idx"]):
        return bad + ["sample indices differ"]
    gx, wx = got["s_xyz"].astype(np.float64), want["s_xyz"].astype(np.float64)
    gc, wc = got["s_cov6"].astype(np.float64), want["s_cov6"].astype(np.float64)
    tr = wc[:, 0] + wc[:, 3] + wc[:, 5]
    scale = np.maximum(tr, 1e-3 * np.median(tr))
    rows = {"position": np.abs(gx - wx).max(1) / np.sqrt(scale), "covariance (6 entries)": np.abs(gc - wc).max(1) / scale}
    for f, rms in (("color", float(want["g_rms_color"])), ("sh", float(want["g_rms_sh"])), ("opacity", float(want["g_rms_opacity"])),
                   ("weight", 1.0)):
        g_, w_ = got["s_" + f].astype(np.float64).reshape(len(wx), -1), want["s_" + f].astype(np.float64).reshape(len(wx), -1)
        rows[f] = np.abs(g_ - w_).max(1) / np.maximum(np.abs(w_).max(1), 5e-2 * rms)
    for name, err in rows.items():
        i = int(np.argmax(err))
        if not err[i] <= tol:
            bad.append(f"sampled row {int(want['sample_idx'][i])}: {name} off by {err[i]:.3e} of the row's own scale")
    return bad
