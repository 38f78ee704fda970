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


# --- the anisotropic ("surfel") workload -------------------------------------------------------------------------
# Trained 3DGS scenes are dominated by flat discs lying on surfaces and by needles along edges, with covariance
# condition numbers of 1e2 ... 1e5, and neighbouring splats share their orientation.  shape="aniso" draws that:
#   60 % discs    the local z axis squashed by 10^U(-2.5, -1)   (sigma ratio 10 ... 316)
#   15 % needles  the local y and z axes squashed by 10^U(-1.5, -0.7) (sigma ratio 5 ... 32)
#   25 % as the isotropic recipe
# and orients the local z axis along a smooth unit normal field n(x) (three plane waves of wavelength ANISO_WAVELEN scene
# units, + N(0, 0.05) jitter), so that nearby discs are nearly coplanar -- random orientations would make every KL
# divergence huge and nothing would merge.  Same sigma_max distribution as the isotropic recipe: the search radii, and
# with them the candidates per parent, stay comparable.
ANISO_WAVELEN = 4.0
_ANISO_DIRS = np.array([[0.8, 0.36, 0.48], [-0.28, 0.9, 0.32], [0.1, -0.5, 0.86]])
_ANISO_PHASE = np.array([0.3, 1.7, 4.1])
_ANISO_AMP = np.array([1.0, 0.8, 0.6])


def _aniso_frames(xyz: np.ndarray, jitter: np.ndarray, psi: np.ndarray, needle: np.ndarray) -> np.ndarray:
    """Rotation matrices (n,3,3), columns = the splats' local x, y, z axes: z along the normal field, x along a coherent
    tangent (needles) or turned by psi about the normal (everything else)."""
    w = (2.0 * np.pi / ANISO_WAVELEN) * (_ANISO_DIRS / np.linalg.norm(_ANISO_DIRS, axis=1, keepdims=True))
    ph = xyz.astype(np.float64) @ w.T + _ANISO_PHASE                       # (n, 3 waves)
    nrm = (np.cos(ph) * _ANISO_AMP) @ w + np.array([0.0, 0.0, 0.35]) + jitter
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    e = np.where((np.abs(nrm[:, 0]) > 0.9)[:, None], np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    t1 = np.cross(nrm, e)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    c, sn = np.cos(psi)[:, None], np.sin(psi)[:, None]
    c = np.where(needle[:, None], 1.0, c)
    sn = np.where(needle[:, None], 0.0, sn)
    x = c * t1 + sn * t2
    y = -sn * t1 + c * t2
    return np.stack([x, y, nrm], 2)


# --- the clustered ("large scene") workload ------------------------------------------------------------------------
# What the reference's README complains about (README.md:113, "for larger scenes the HEM downsampler becomes extremely slow"):
# strongly non-uniform density and a few huge splats.  shape="clustered" draws
#   60 % of the splats in CLUSTER_CLUMPS Gaussian clumps whose peak density is 30 ... 100 x the background's; their splats are
#        smaller by the cube root of that factor (as trained scenes have it: small splats where there are many), so a parent of a
#        clump merges about as many children as one of the background -- but the uniform grid's cells, sized for the average
#        density, hold 30 ... 100 x more components there: long rows, heavy parents, crowded sum buckets;
#   40 % a sparse uniform background (the isotropic recipe);
#   CLUSTER_GIANTS background splats with 40 x the standard deviations (search radii of the size of the scene: the reference's grid
#        cell is the largest parent radius, mixture.cpp:92-99, which is the pathology), and CLUSTER_OUTLIERS far outliers at 20 ... 60 h.
CLUSTER_CLUMPS, CLUSTER_GIANTS, CLUSTER_OUTLIERS = 40, 6, 12


def _clustered_layout(n: int, h: float, rng):
    """Host-side plan shared by the numpy and the torch generator: clump sizes / centres / sigmas / scale factors, giants, outliers."""
    n_cl = int(0.6 * n)
    wts = rng.uniform(0.5, 1.5, CLUSTER_CLUMPS)
    sizes = np.floor(wts / wts.sum() * n_cl).astype(np.int64)
    sizes[0] += n_cl - sizes.sum()
    centres = rng.uniform(-0.8 * h, 0.8 * h, (CLUSTER_CLUMPS, 3))
    f = rng.uniform(30.0, 100.0, CLUSTER_CLUMPS)                          # peak density over the background's
    rho_bg = 0.4 * n / (2.0 * h) ** 3
    sigma = (sizes / ((2.0 * np.pi) ** 1.5 * f * rho_bg)) ** (1.0 / 3.0)
    shrink = f ** (-1.0 / 3.0)
    n_bg = n - n_cl
    giants = n_cl + rng.choice(n_bg, min(CLUSTER_GIANTS, n_bg), replace=False)        # indices into the background part
    outl = n_cl + rng.choice(n_bg, min(CLUSTER_OUTLIERS, n_bg), replace=False)
    far = rng.uniform(20.0, 60.0, (len(outl), 1)) * h * rng.choice([-1.0, 1.0], (len(outl), 3))
    return dict(sizes=sizes, centres=centres, sigma=sigma, shrink=shrink, giants=giants, outliers=outl, far=far, n_cl=n_cl)


def make_cloud(n: int, seed: int = 0, h: float | None = None, sh_degree: int = 3,
               chunk: int = 1 << 20, shape: str = "iso") -> dict:
    """Return the five level-0 arrays the HEM boundary takes, as float32 numpy arrays.  shape: "iso" (SURVEY 8(d)), "aniso" (discs
    and needles on a smooth orientation field) or "clustered" (dense clumps, a sparse background, giants and far outliers), see above."""
    if h is None:
        h = half_extent(n)
    F = 3 * ((sh_degree + 1) ** 2 - 1)
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-h, h, (n, 3)).astype(np.float32)
    s = np.exp(rng.normal(-2.5, 0.5, (n, 3))).astype(np.float32)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    frames = None
    if shape == "aniso":
        arng = np.random.default_rng(seed + 104729)          # its own stream: the "iso" draws above stay what they were
        u = arng.random(n)
        disc, needle = u < 0.60, (u >= 0.60) & (u < 0.75)
        s = s.astype(np.float64)
        s[:, 2] *= np.where(disc, 10.0 ** arng.uniform(-2.5, -1.0, n), 1.0)
        f = 10.0 ** arng.uniform(-1.5, -0.7, n)
        s[:, 1] *= np.where(needle, f, 1.0)
        s[:, 2] *= np.where(needle, f, 1.0)
        s = s.astype(np.float32)
        frames = (arng.normal(0, 0.05, (n, 3)), arng.uniform(0, 2 * np.pi, n), needle)
    elif shape == "clustered":
        crng = np.random.default_rng(seed + 15485863)       # its own stream
        lay = _clustered_layout(n, h, crng)
        which = np.repeat(np.arange(CLUSTER_CLUMPS), lay["sizes"])
        xyz[:lay["n_cl"]] = (lay["centres"][which] + crng.normal(0, 1, (lay["n_cl"], 3)) * lay["sigma"][which, None]).astype(np.float32)
        s[:lay["n_cl"]] *= lay["shrink"][which, None].astype(np.float32)
        s[lay["giants"]] *= np.float32(40.0)
        xyz[lay["outliers"]] = lay["far"].astype(np.float32)
        perm = crng.permutation(n)                           # clumps, background, giants and outliers mixed in the input order
        xyz, s, q = xyz[perm], s[perm], q[perm]
    elif shape != "iso":
        raise ValueError(f"unknown cloud shape {shape!r}")
    cov6 = np.empty((n, 6), dtype=np.float32)
    for a in range(0, n, chunk):           # chunked: the (n,3,3) float64 temporaries are large at 5M
        b = min(n, a + chunk)
        R = _quat_to_rot(q[a:b]) if frames is None else _aniso_frames(xyz[a:b], frames[0][a:b], frames[1][a:b], frames[2][a:b])
        L = R * s[a:b, None, :].astype(np.float64)
        C = (L @ L.transpose(0, 2, 1)).astype(np.float32)
        cov6[a:b] = C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
    col = rng.normal(0, 0.5, (n, 3)).astype(np.float32)
    op = rng.normal(0, 2.0, (n,)).astype(np.float32)
    sh = rng.normal(0, 0.1, (n, F)).astype(np.float32)
    return {"xyz": xyz, "color": col, "opacity": op, "cov6": cov6, "sh": sh,
            "sh_degree": sh_degree, "h": float(h), "shape": shape}


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


def make_cloud_torch(n: int, seed: int = 0, device="cuda:0", sh_degree: int = 3, h: float | None = None, shape: str = "iso"):
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
    if shape == "aniso":                        # the recipe of make_cloud(shape="aniso"), in float64 on the device
        ag = torch.Generator(device=device).manual_seed(seed + 104729)
        u = torch.rand((n,), device=device, generator=ag)
        disc, needle = u < 0.60, (u >= 0.60) & (u < 0.75)
        one = torch.ones((), device=device)
        s = s.clone()
        s[:, 2] *= torch.where(disc, 10.0 ** (torch.rand((n,), device=device, generator=ag) * 1.5 - 2.5), one)
        f = 10.0 ** (torch.rand((n,), device=device, generator=ag) * 0.8 - 1.5)
        s[:, 1] *= torch.where(needle, f, one)
        s[:, 2] *= torch.where(needle, f, one)
        jit = torch.randn((n, 3), device=device, generator=ag, dtype=torch.float64) * 0.05
        psi = torch.rand((n,), device=device, generator=ag, dtype=torch.float64) * (2 * np.pi)
        dirs = torch.as_tensor(_ANISO_DIRS / np.linalg.norm(_ANISO_DIRS, axis=1, keepdims=True), device=device)
        wv = (2.0 * np.pi / ANISO_WAVELEN) * dirs
        ph = xyz.double() @ wv.T + torch.as_tensor(_ANISO_PHASE, device=device)
        nrm = (torch.cos(ph) * torch.as_tensor(_ANISO_AMP, device=device)) @ wv + torch.tensor([0.0, 0.0, 0.35], device=device, dtype=torch.float64) + jit
        nrm = torch.nn.functional.normalize(nrm, dim=1)
        ex = torch.tensor([1.0, 0.0, 0.0], device=device, dtype=torch.float64).expand(n, 3)
        ey = torch.tensor([0.0, 1.0, 0.0], device=device, dtype=torch.float64).expand(n, 3)
        e = torch.where((nrm[:, 0].abs() > 0.9)[:, None], ey, ex)
        t1 = torch.nn.functional.normalize(torch.linalg.cross(nrm, e), dim=1)
        t2 = torch.linalg.cross(nrm, t1)
        c = torch.where(needle, torch.ones_like(psi), torch.cos(psi))[:, None]
        sn = torch.where(needle, torch.zeros_like(psi), torch.sin(psi))[:, None]
        R = torch.stack([c * t1 + sn * t2, -sn * t1 + c * t2, nrm], 2).float()       # columns = local x, y, z
        del jit, psi, ph, nrm, e, t1, t2, c, sn
    elif shape not in ("iso", "clustered"):
        raise ValueError(f"unknown cloud shape {shape!r}")
    else:
        if shape == "clustered":                # the recipe of make_cloud(shape="clustered"): the plan on the host, the draws on the device
            lay = _clustered_layout(n, h, np.random.default_rng(seed + 15485863))
            cg = torch.Generator(device=device).manual_seed(seed + 15485863)
            t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=device)
            which = torch.repeat_interleave(torch.arange(CLUSTER_CLUMPS, device=device), t(lay["sizes"], torch.int64))
            n_cl = lay["n_cl"]
            xyz[:n_cl] = t(lay["centres"])[which] + torch.randn((n_cl, 3), device=device, generator=cg) * t(lay["sigma"])[which, None]
            s[:n_cl] *= t(lay["shrink"])[which, None]
            s[t(lay["giants"], torch.int64)] *= 40.0
            xyz[t(lay["outliers"], torch.int64)] = t(lay["far"])
            perm = torch.randperm(n, device=device, generator=cg)
            xyz, s, q = xyz[perm].contiguous(), s[perm].contiguous(), q[perm].contiguous()
            del which, perm
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
    return {"xyz": xyz, "color": col, "opacity": op, "cov6": cov6, "sh": sh, "sh_degree": sh_degree, "h": float(h), "shape": shape}


def block_box(rank: int, world: int, h: float):
    """(lo, hi) corners of block ``rank`` of ``world``: the cube [-h, h]^3 cut into ``parallel.block_dims(world)`` equal boxes
    (px along x, py along y, pz along z; rank = (ix * py + iy) * pz + iz, the numbering of ``parallel.block_of``)."""
    from .parallel import block_dims
    px, py, pz = block_dims(world)
    ix, iy, iz = rank // (py * pz), (rank // pz) % py, rank % pz
    lo = np.array([-h + 2 * h * ix / px, -h + 2 * h * iy / py, -h + 2 * h * iz / pz])
    hi = np.array([-h + 2 * h * (ix + 1) / px, -h + 2 * h * (iy + 1) / py, -h + 2 * h * (iz + 1) / pz])
    return lo, hi


def make_block_cloud_torch(n_global: int, rank: int, world: int, seed: int = 0, device="cuda:0", sh_degree: int = 3, shape: str = "iso"):
    """ONE rank's block of a cloud of ``n_global`` splats that is never materialised as a whole (BASELINE config 5 at 40 M: every
    rank used to draw and argsort the full cloud before cutting its block out).  The cloud is DEFINED block by block: block r holds
    the global indices ``parallel.shard_range(n_global, r, world)`` (ascending, contiguous), drawn with ``make_cloud_torch``'s recipe
    from the seed ``seed * 1000003 + r`` uniformly inside ``block_box(r, world, half_extent(n_global))`` -- the same density and the
    same equal-count, equal-volume cut as ``parallel.block_of`` gives on a uniform cloud.  Returns (cloud dict, gid int32 tensor)."""
    import torch
    from .parallel import shard_range
    h = half_extent(n_global)
    lo_i, hi_i = shard_range(n_global, rank, world)
    c = make_cloud_torch(hi_i - lo_i, seed=seed * 1000003 + rank, device=device, sh_degree=sh_degree, h=1.0, shape="iso")
    lo, hi = block_box(rank, world, h)
    ctr = torch.as_tensor((lo + hi) * 0.5, dtype=torch.float32, device=c["xyz"].device)
    half = torch.as_tensor((hi - lo) * 0.5, dtype=torch.float32, device=c["xyz"].device)
    c["xyz"] = (c["xyz"] * half + ctr).contiguous()
    if shape != "iso":
        raise ValueError("make_block_cloud_torch draws the isotropic recipe only")
    c["h"] = float(h)
    gid = torch.arange(lo_i, hi_i, dtype=torch.int32, device=c["xyz"].device)
    return c, gid


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
