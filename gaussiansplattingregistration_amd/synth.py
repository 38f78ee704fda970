"""Synthetic Gaussian-splat clouds for tests and benchmarks (own code).

The reference ships no sample data (its ``inputs/`` holds a ``.gitkeep`` only), so every test and
bench in this repo runs on clouds drawn here.  The recipe is fixed so that fixture generators,
the CPU oracle and the GPU path all see the same bytes:

    rng  = numpy.random.default_rng(seed)
    xyz  ~ U(-h, h)^3                       float32
    s    = exp(N(-2.5, 0.5))  (n,3)         float32     per-axis standard deviations
    q    ~ N(0,1)^4 normalised (w,x,y,z)    float64
    cov  = R diag(s^2) R^T   in float64 -> float32, packed (xx, xy, xz, yy, yz, zz)
    col  ~ N(0, 0.5)  (n,3)                 float32     SH DC term
    op   ~ N(0, 2.0)  (n,)                  float32     RAW (pre-sigmoid) opacity
    sh   ~ N(0, 0.1)  (n,F)                 float32     SH rest, F = 3*((deg+1)^2 - 1)

``h = 1.5 * (n / 50000)^(1/3)`` keeps the density constant (about 1850 splats per unit volume).
The packing order of ``cov`` is the one ``GaussianModel.get_covariance`` produces in the reference
(``src/utils/general_utils.py:20-29``) and ``smat3`` consumes (``src/cpp_ext/include/vec.hpp:458``).
"""
from __future__ import annotations

import numpy as np

__all__ = ["half_extent", "make_cloud", "rigid_transform", "make_pair", "apply_rigid"]


def half_extent(n: int) -> float:
    """Box half-extent that keeps the splat density of the 50k-splat probe case."""
    return 1.5 * (n / 50000.0) ** (1.0 / 3.0)


def _quat_to_rot(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((q.shape[0], 3, 3), dtype=np.float64)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def make_cloud(n: int, seed: int = 0, h: float | None = None, sh_degree: int = 3,
               chunk: int = 1 << 20) -> dict:
    """Return the five level-0 arrays the HEM boundary takes, as float32 numpy arrays."""
    if h is None:
        h = half_extent(n)
    F = 3 * ((sh_degree + 1) ** 2 - 1)
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-h, h, (n, 3)).astype(np.float32)
    s = np.exp(rng.normal(-2.5, 0.5, (n, 3))).astype(np.float32)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    cov6 = np.empty((n, 6), dtype=np.float32)
    for a in range(0, n, chunk):           # chunked: the (n,3,3) float64 temporaries are large at 5M
        b = min(n, a + chunk)
        R = _quat_to_rot(q[a:b])
        L = R * s[a:b, None, :].astype(np.float64)
        C = (L @ L.transpose(0, 2, 1)).astype(np.float32)
        cov6[a:b] = C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
    col = rng.normal(0, 0.5, (n, 3)).astype(np.float32)
    op = rng.normal(0, 2.0, (n,)).astype(np.float32)
    sh = rng.normal(0, 0.1, (n, F)).astype(np.float32)
    return {"xyz": xyz, "color": col, "opacity": op, "cov6": cov6, "sh": sh,
            "sh_degree": sh_degree, "h": float(h)}


def rigid_transform(angle_deg: float = 5.0, axis=(1.0, 1.0, 1.0), translation=(0.0, 0.0, 0.0)) -> np.ndarray:
    """4x4 float64 rigid transform: Rodrigues rotation about ``axis`` then ``translation``."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    th = np.deg2rad(angle_deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = np.asarray(translation, dtype=np.float64)
    return T


def apply_rigid(cloud: dict, T: np.ndarray) -> dict:
    """Rigidly move a cloud: xyz' = R xyz + t, cov' = R cov R^T (float64 maths, float32 storage)."""
    R, t = T[:3, :3], T[:3, 3]
    out = dict(cloud)
    out["xyz"] = (cloud["xyz"].astype(np.float64) @ R.T + t).astype(np.float32)
    c = cloud["cov6"].astype(np.float64)
    C = np.empty((c.shape[0], 3, 3))
    C[:, 0, 0], C[:, 0, 1], C[:, 0, 2] = c[:, 0], c[:, 1], c[:, 2]
    C[:, 1, 0], C[:, 1, 1], C[:, 1, 2] = c[:, 1], c[:, 3], c[:, 4]
    C[:, 2, 0], C[:, 2, 1], C[:, 2, 2] = c[:, 2], c[:, 4], c[:, 5]
    C = R @ C @ R.T
    out["cov6"] = C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]].astype(np.float32)
    return out


def make_pair(n: int, seed: int = 0, sh_degree: int = 3, jitter: float = 0.002,
              angle_deg: float = 5.0):
    """A registration pair with a known answer.

    ``B`` is cloud ``A``; the *source* is ``A`` moved by ``inv(T_gt)`` plus N(0, jitter) position
    noise, so that registering source -> target has the unique solution ``T_gt`` (SURVEY.md 8(d),
    the "ICP-only" variant).  Returns ``(source, target, T_gt)``.
    """
    target = make_cloud(n, seed=seed, sh_degree=sh_degree)
    h = target["h"]
    T_gt = rigid_transform(angle_deg, (1, 1, 1), 0.05 * h * np.array([1.0, -1.0, 0.5]))
    source = apply_rigid(target, np.linalg.inv(T_gt))
    if jitter > 0:
        rng = np.random.default_rng(seed + 7919)
        source["xyz"] = (source["xyz"] + rng.normal(0, jitter, source["xyz"].shape)).astype(np.float32)
    return source, target, T_gt


def make_cloud_torch(n: int, seed: int = 0, device="cuda:0", sh_degree: int = 3, h: float | None = None):
    """Same distributions as ``make_cloud`` drawn directly on the GPU with torch's generator (for the
    5M-splat benchmark / full-size tests: no host staging).  Not the same stream as the NumPy recipe."""
    import torch
    if h is None:
        h = half_extent(n)
    F = 3 * ((sh_degree + 1) ** 2 - 1)
    g = torch.Generator(device=device).manual_seed(seed)
    xyz = (torch.rand((n, 3), device=device, generator=g) * 2 - 1) * h
    s = torch.exp(torch.randn((n, 3), device=device, generator=g) * 0.5 - 2.5)
    q = torch.nn.functional.normalize(torch.randn((n, 4), device=device, generator=g), dim=1)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).view(n, 3, 3)
    # Sigma = R diag(s^2) R^T written out element-wise: a batched matmul (torch.bmm) with more than 2^24 batches faults
    # on this ROCm build, and the large-cloud runs (20 M, 40 M splats) go past that
    s2 = s * s
    cc = lambda a, b: (R[:, a, :] * R[:, b, :] * s2).sum(1)
    cov6 = torch.stack([cc(0, 0), cc(0, 1), cc(0, 2), cc(1, 1), cc(1, 2), cc(2, 2)], 1).contiguous()
    del R, s2
    col = torch.randn((n, 3), device=device, generator=g) * 0.5
    op = torch.randn((n,), device=device, generator=g) * 2.0
    sh = torch.randn((n, F), device=device, generator=g) * 0.1
    return {"xyz": xyz, "color": col, "opacity": op, "cov6": cov6, "sh": sh, "sh_degree": sh_degree, "h": float(h)}


def apply_rigid_torch(cloud: dict, T):
    """Rigid motion of a device cloud (float64 maths, float32 storage)."""
    import torch
    Tt = torch.as_tensor(np.asarray(T), dtype=torch.float64, device=cloud["xyz"].device)
    R, t = Tt[:3, :3], Tt[:3, 3]
    out = dict(cloud)
    out["xyz"] = (cloud["xyz"].double() @ R.T + t).float()
    c = cloud["cov6"].double()
    C = torch.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], 1)
    n = C.shape[0]
    D = (C.reshape(n * 3, 3) @ R.T).reshape(n, 3, 3)                       # C R^T as ONE (3n x 3) GEMM (no batched matmul:
    C = (D.transpose(1, 2).reshape(n * 3, 3) @ R.T).reshape(n, 3, 3)       # torch.bmm faults past 2^24 batches here), then R (.)
    C = C.transpose(1, 2)
    out["cov6"] = torch.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], 1).float().contiguous()
    return out
