"""Synthetic scene for the plane-fitting tests (own code): three planar patches with noisy normals + clutter."""
import numpy as np


def scene(seed):
    """Three planar patches with noisy normals + clutter."""
    rng = np.random.default_rng(seed)
    pts, nrm = [], []
    for n0, off, m in (((0, 0, 1), 0.0, 6000), ((1, 0, 0.2), 0.8, 3500), ((0.1, 1, 0), -0.6, 2500)):
        n0 = np.asarray(n0, float); n0 /= np.linalg.norm(n0)
        u = np.cross(n0, [0.3, 0.5, 0.8]); u /= np.linalg.norm(u)
        v = np.cross(n0, u)
        ab = rng.uniform(-1.5, 1.5, (m, 2))
        p = ab[:, :1] * u + ab[:, 1:] * v + n0 * off + rng.normal(0, 0.004, (m, 1)) * n0
        nn = n0 + rng.normal(0, 0.05, (m, 3))
        pts.append(p); nrm.append(nn / np.linalg.norm(nn, axis=1, keepdims=True) * rng.choice([-1, 1], (m, 1)))
    m = 3000
    pts.append(rng.uniform(-1.5, 1.5, (m, 3)))
    nn = rng.normal(size=(m, 3))
    nrm.append(nn / np.linalg.norm(nn, axis=1, keepdims=True))
    return np.concatenate(pts), np.concatenate(nrm)
