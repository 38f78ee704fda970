"""ctypes front end of the CPU oracle (oracle/libgsr_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module -- as the checker or the reported CPU baseline, never as the thing measured or shipped.
The product package ``gaussiansplattingregistration_amd`` does not import it.

HEM half: parity pinned against the reference's compiled cpp_ext (see hem_oracle.cpp header).
ICP half: parity UNPINNED (Open3D 0.16.0 absent; see icp_oracle.cpp header).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("GSR_ORACLE_LIB") or os.path.join(_HERE, "libgsr_oracle.so")      # GSR_ORACLE_LIB: the sanitizer build
_lib = None

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile oracle/libgsr_oracle.so (g++, a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("hem_oracle.cpp", "icp_oracle.cpp", "glibc_rand.h", "oracle_api.h")]
    if os.environ.get("GSR_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.gsr_oracle_hem_create.restype = vp
    L.gsr_oracle_hem_create.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int32,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint64]
    L.gsr_oracle_hem_destroy.argtypes = [vp]
    L.gsr_oracle_hem_set_parent_mask.argtypes = [vp, _u8p]
    L.gsr_oracle_hem_set_weights.argtypes = [vp, _f32p]
    L.gsr_oracle_hem_level.restype = C.c_int64
    L.gsr_oracle_hem_level.argtypes = [vp, C.c_int32]
    L.gsr_oracle_hem_num_levels.argtypes = [vp]
    L.gsr_oracle_hem_set_fast_search.argtypes = [vp, C.c_int32]
    L.gsr_oracle_hem_used_fast_search.argtypes = [vp]
    L.gsr_oracle_hem_level_size.restype = C.c_int64
    L.gsr_oracle_hem_level_size.argtypes = [vp, C.c_int32]
    L.gsr_oracle_hem_get_level.argtypes = [vp, C.c_int32] + [vp] * 7
    L.gsr_oracle_hem_stats.argtypes = [vp, _i64p]
    L.gsr_oracle_hem_margins.argtypes = [vp, _f64p]
    L.gsr_oracle_hem_phase_times.argtypes = [vp, _f64p]
    L.gsr_oracle_rand_stream.argtypes = [C.c_uint32, C.c_uint64, C.c_int64, _u32p]
    L.gsr_oracle_parent_flags.argtypes = [C.c_uint32, C.c_uint64, C.c_float, C.c_int64, _u8p]
    L.gsr_oracle_eigenvalues.argtypes = [_f32p, C.c_int64, _f32p]
    L.gsr_oracle_det.argtypes = [_f32p, C.c_int64, _f32p]
    L.gsr_oracle_kld.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64, _f32p]
    L.gsr_oracle_logf.restype = C.c_float
    L.gsr_oracle_logf.argtypes = [C.c_float]
    L.gsr_oracle_icp.restype = C.c_int32
    L.gsr_oracle_icp.argtypes = [_f64p, C.c_int64, _f64p, vp, C.c_int64, _f64p, C.c_int32, C.c_int32, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _f64p,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.gsr_oracle_icp_correspond.argtypes = [_f64p, C.c_int64, _f64p, C.c_int64, _f64p, C.c_double, C.c_int32, _i64p, _f64p]
    L.gsr_oracle_normals_from_cov.argtypes = [_f64p, C.c_int64, _f64p]
    _lib = L
    return L


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class HemOracle:
    """One mixture hierarchy on the CPU oracle (mirrors hem::Mixture, mixture.hpp:47-76)."""

    def __init__(self, xyz, color, cov6, opacity, sh, rho=3.0, delta=3.0, kappa=2.5, tau=1.0,
                 rng_seed=1, rng_skip=0):
        L = lib()
        xyz = _c(xyz, np.float32).reshape(-1, 3)
        n = xyz.shape[0]
        sh = _c(sh, np.float32).reshape(n, -1) if n else _c(sh, np.float32).reshape(0, 0)
        self.F = sh.shape[1]
        self._L = L
        self._h = L.gsr_oracle_hem_create(xyz, _c(color, np.float32).reshape(n, 3), _c(cov6, np.float32).reshape(n, 6),
                                          _c(opacity, np.float32).reshape(n), sh if sh.size else np.zeros(1, np.float32),
                                          n, self.F, rho, delta, kappa, tau, rng_seed, rng_skip)
        if not self._h:
            raise RuntimeError("gsr_oracle_hem_create failed")

    def close(self):
        if self._h:
            self._L.gsr_oracle_hem_destroy(self._h)
            self._h = None

    __del__ = close

    def set_parent_mask(self, mask):
        self._L.gsr_oracle_hem_set_parent_mask(self._h, _c(mask, np.uint8))

    def set_weights(self, w):
        self._L.gsr_oracle_hem_set_weights(self._h, _c(w, np.float32))

    def set_fast_search(self, on=True):
        """The reference's result lists (same members, same order) through a finer grid: for clouds whose 27-cell scans would take
        hours (cell = the largest parent radius, mixture.cpp:92-99).  ``used_fast_search`` says whether the last level took it."""
        self._L.gsr_oracle_hem_set_fast_search(self._h, 1 if on else 0)

    @property
    def used_fast_search(self) -> bool:
        return bool(self._L.gsr_oracle_hem_used_fast_search(self._h))

    def run_level(self, threads=0) -> int:
        n = self._L.gsr_oracle_hem_level(self._h, threads)
        if n < 0:
            raise RuntimeError("oracle level failed")
        return int(n)

    @property
    def num_levels(self) -> int:
        return self._L.gsr_oracle_hem_num_levels(self._h)

    def level(self, k: int) -> dict:
        n = int(self._L.gsr_oracle_hem_level_size(self._h, k))
        out = {"xyz": np.empty((n, 3), np.float32), "color": np.empty((n, 3), np.float32),
               "cov6": np.empty((n, 6), np.float32), "opacity": np.empty(n, np.float32),
               "sh": np.empty((n, self.F), np.float32), "weight": np.empty(n, np.float32),
               "is_parent": np.empty(n, np.uint8)}
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._L.gsr_oracle_hem_get_level(self._h, k, p(out["xyz"]), p(out["color"]), p(out["cov6"]), p(out["opacity"]),
                                         p(out["sh"]), p(out["weight"]), p(out["is_parent"]))
        return out

    def stats(self) -> dict:
        s = np.zeros(6, np.int64)
        self._L.gsr_oracle_hem_stats(self._h, s)
        m = np.zeros(2, np.float64)
        self._L.gsr_oracle_hem_margins(self._h, m)
        t = np.zeros(5, np.float64)
        self._L.gsr_oracle_hem_phase_times(self._h, t)
        return {"parents": int(s[0]), "pairs": int(s[1]), "orphans": int(s[2]), "dropped": int(s[3]),
                "candidates": int(s[4]), "draws": int(s[5]), "kld_margin": float(m[0]), "color_margin": float(m[1]),
                "t_grid": t[0], "t_select": t[1], "t_likelihood": t[2], "t_mstep": t[3], "t_rest": t[4]}


def hem(cloud: dict, levels: int, rho=3.0, delta=3.0, kappa=2.5, tau=1.0, threads=0, rng_seed=1, rng_skip=0, fast_search=False):
    """MixtureCreator::CreateMixture on the oracle: list of ``levels`` level dicts (level 0 dropped).  ``fast_search``: the same result
    lists through a finer grid (HemOracle.set_fast_search); stats carry ``fast_search`` = whether a level took it."""
    o = HemOracle(cloud["xyz"], cloud["color"], cloud["cov6"], cloud["opacity"], cloud["sh"], rho, delta, kappa, tau,
                  rng_seed, rng_skip)
    o.set_fast_search(fast_search)
    out, st = [], []
    for _ in range(levels):
        o.run_level(threads)
        st.append(dict(o.stats(), fast_search=o.used_fast_search))
        out.append(o.level(o.num_levels - 1))
    o.close()
    return out, st


def rand_stream(n, seed=1, skip=0):
    out = np.empty(n, np.uint32)
    lib().gsr_oracle_rand_stream(seed, skip, n, out)
    return out


def parent_flags(n, rho, seed=1, skip=0):
    out = np.empty(n, np.uint8)
    lib().gsr_oracle_parent_flags(seed, skip, rho, n, out)
    return out


def eigenvalues(cov6):
    c = _c(cov6, np.float32).reshape(-1, 6)
    out = np.empty((c.shape[0], 3), np.float32)
    lib().gsr_oracle_eigenvalues(c, c.shape[0], out)
    return out


def det(cov6):
    c = _c(cov6, np.float32).reshape(-1, 6)
    out = np.empty(c.shape[0], np.float32)
    lib().gsr_oracle_det(c, c.shape[0], out)
    return out


def kld(child_mean, child_cov6, parent_mean, parent_cov6):
    cm = _c(child_mean, np.float32).reshape(-1, 3)
    out = np.empty(cm.shape[0], np.float32)
    lib().gsr_oracle_kld(cm, _c(child_cov6, np.float32).reshape(-1, 6), _c(parent_mean, np.float32).reshape(-1, 3),
                         _c(parent_cov6, np.float32).reshape(-1, 6), cm.shape[0], out)
    return out


def logf(x: float) -> float:
    return float(lib().gsr_oracle_logf(float(x)))


def icp(src, tgt, tgt_normals=None, init=None, kind=0, loss=0, k=0.0, max_corr=1.0, rel_fitness=1e-6, rel_rmse=1e-6,
        max_iter=30, threads=0, want_trace=False):
    """registration_icp on the oracle.  Returns dict(transformation, fitness, inlier_rmse, iterations[, trace])."""
    src = _c(src, np.float64).reshape(-1, 3)
    tgt = _c(tgt, np.float64).reshape(-1, 3)
    nrm = None if tgt_normals is None else _c(tgt_normals, np.float64).reshape(-1, 3)
    init = np.eye(4) if init is None else _c(init, np.float64).reshape(4, 4)
    T = np.empty((4, 4), np.float64)
    fit, rmse = C.c_double(0), C.c_double(0)
    trace = np.zeros((max_iter + 1, 18), np.float64) if want_trace else None
    it = lib().gsr_oracle_icp(src, src.shape[0], tgt, None if nrm is None else nrm.ctypes.data_as(C.c_void_p),
                              tgt.shape[0], init, kind, loss, k, max_corr, rel_fitness, rel_rmse, max_iter, threads,
                              T, C.byref(fit), C.byref(rmse), None if trace is None else trace.ctypes.data_as(C.c_void_p))
    if it < 0:
        raise RuntimeError({-1: "max_correspondence_distance must be > 0",
                            -2: "point-to-plane ICP requires target normals",
                            -3: "empty point cloud"}.get(it, f"icp oracle error {it}"))
    out = {"transformation": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iterations": int(it)}
    if want_trace:
        out["trace"] = trace[: it + 1]
    return out


def gicp(src, src_cov, tgt, tgt_cov, init=None, loss=0, k=0.0, max_corr=1.0, rel_fitness=1e-6, rel_rmse=1e-6, max_iter=30,
         threads=0):
    """registration_generalized_icp on the oracle (covariances (N,3,3) float64, used as given)."""
    src = _c(src, np.float64).reshape(-1, 3)
    tgt = _c(tgt, np.float64).reshape(-1, 3)
    sc = _c(src_cov, np.float64).reshape(-1, 9)
    tc = _c(tgt_cov, np.float64).reshape(-1, 9)
    init = np.eye(4) if init is None else _c(init, np.float64).reshape(4, 4)
    T = np.empty((4, 4), np.float64)
    fit, rmse = C.c_double(0), C.c_double(0)
    fn = lib().gsr_oracle_gicp
    fn.restype = C.c_int32
    P = C.c_void_p
    fn.argtypes = [P, P, C.c_int64, P, P, C.c_int64, P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                   C.c_int32, P, P, P]
    it = fn(src.ctypes.data, sc.ctypes.data, src.shape[0], tgt.ctypes.data, tc.ctypes.data, tgt.shape[0], init.ctypes.data, loss, k,
            max_corr, rel_fitness, rel_rmse, max_iter, threads, T.ctypes.data, C.addressof(fit), C.addressof(rmse))
    if it < 0:
        raise RuntimeError({-1: "max_correspondence_distance must be > 0", -2: "generalized ICP requires covariances",
                            -3: "empty point cloud"}.get(it, f"gicp oracle error {it}"))
    return {"transformation": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iterations": int(it)}


def voxel_down_sample(xyz, voxel_size, color=None, cov=None):
    """Open3D ``voxel_down_sample`` on float64 arrays; voxels in ascending (ix, iy, iz) order.  -> (xyz, color, cov)."""
    x = _c(xyz, np.float64).reshape(-1, 3)
    col = None if color is None else _c(color, np.float64).reshape(-1, 3)
    cv = None if cov is None else _c(cov, np.float64).reshape(-1, 9)
    fn = lib().gsr_oracle_voxel_down_sample
    fn.restype = C.c_int64
    P = C.c_void_p
    fn.argtypes = [P, P, P, C.c_int64, C.c_double, P, P, P]
    pc = None if col is None else col.ctypes.data
    pv = None if cv is None else cv.ctypes.data
    v = fn(x.ctypes.data, pc, pv, x.shape[0], float(voxel_size), None, None, None)
    if v < 0:
        raise RuntimeError("[VoxelDownSample] voxel_size <= 0.")
    ox = np.empty((v, 3), np.float64)
    oc = None if col is None else np.empty((v, 3), np.float64)
    ov = None if cv is None else np.empty((v, 9), np.float64)
    fn(x.ctypes.data, pc, pv, x.shape[0], float(voxel_size), ox.ctypes.data, None if oc is None else oc.ctypes.data,
       None if ov is None else ov.ctypes.data)
    return ox, oc, None if ov is None else ov.reshape(-1, 3, 3)


def color_gradient(tgt, tgt_normals, tgt_colors, radius, max_nn=30, threads=0):
    t = _c(tgt, np.float64).reshape(-1, 3)
    n = _c(tgt_normals, np.float64).reshape(-1, 3)
    c = _c(tgt_colors, np.float64).reshape(-1, 3)
    out = np.empty_like(t)
    fn = lib().gsr_oracle_color_gradient
    fn.restype = None
    P = C.c_void_p
    fn.argtypes = [P, P, P, C.c_int64, C.c_double, C.c_int32, C.c_int32, P]
    fn(t.ctypes.data, n.ctypes.data, c.ctypes.data, t.shape[0], float(radius), int(max_nn), int(threads), out.ctypes.data)
    return out


def colored_icp(src, src_colors, tgt, tgt_normals, tgt_colors, init=None, loss=0, k=0.0, lambda_geometric=0.968, max_corr=1.0,
                rel_fitness=1e-6, rel_rmse=1e-6, max_iter=30, threads=0):
    """registration_colored_icp on the oracle."""
    s = _c(src, np.float64).reshape(-1, 3)
    sc = _c(src_colors, np.float64).reshape(-1, 3)
    t = _c(tgt, np.float64).reshape(-1, 3)
    tn = _c(tgt_normals, np.float64).reshape(-1, 3)
    tc = _c(tgt_colors, np.float64).reshape(-1, 3)
    init = np.eye(4) if init is None else _c(init, np.float64).reshape(4, 4)
    T = np.empty((4, 4), np.float64)
    fit, rmse = C.c_double(0), C.c_double(0)
    fn = lib().gsr_oracle_colored_icp
    fn.restype = C.c_int32
    P = C.c_void_p
    fn.argtypes = [P, P, C.c_int64, P, P, P, C.c_int64, P, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                   C.c_int32, C.c_int32, P, P, P]
    it = fn(s.ctypes.data, sc.ctypes.data, s.shape[0], t.ctypes.data, tn.ctypes.data, tc.ctypes.data, t.shape[0], init.ctypes.data,
            loss, k, lambda_geometric, max_corr, rel_fitness, rel_rmse, max_iter, threads, T.ctypes.data, C.addressof(fit),
            C.addressof(rmse))
    if it < 0:
        raise RuntimeError({-1: "max_correspondence_distance must be > 0", -2: "colored ICP requires target normals",
                            -3: "empty point cloud", -4: "colored ICP requires colours"}.get(it, f"colored icp oracle error {it}"))
    return {"transformation": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iterations": int(it)}


def icp_correspond(src, tgt, T, max_corr, threads=0):
    src = _c(src, np.float64).reshape(-1, 3)
    tgt = _c(tgt, np.float64).reshape(-1, 3)
    idx = np.empty(src.shape[0], np.int64)
    d2 = np.empty(src.shape[0], np.float64)
    lib().gsr_oracle_icp_correspond(src, src.shape[0], tgt, tgt.shape[0], _c(T, np.float64).reshape(4, 4), max_corr,
                                    threads, idx, d2)
    return idx, d2


def normals_from_cov(cov3x3):
    c = _c(cov3x3, np.float64).reshape(-1, 3, 3)
    out = np.empty((c.shape[0], 3), np.float64)
    lib().gsr_oracle_normals_from_cov(c, c.shape[0], out)
    return out


def normals_knn(points, knn=30, threads=0):
    """Open3D ``estimate_normals()`` with its default ``KDTreeSearchParamKNN(30)`` on a cloud without covariances."""
    p = _c(points, np.float64).reshape(-1, 3)
    out = np.empty_like(p)
    fn = lib().gsr_oracle_normals_knn
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    fn(p.ctypes.data, p.shape[0], int(knn), int(threads), out.ctypes.data)
    return out


def decompose_reference(cov6):
    """CPU restatement (numpy, float64 eigh) of GaussianModel.decompose_covariance_matrix + matrices_to_quaternions
    (reference src/models/gaussian_model.py:242-265, src/utils/general_utils.py:94-100), pinned against
    tests/golden/from_mixture.npz (generated by running the reference's own functions, tests/golden/make_golden_from_mixture.py).
    Eigenvector sign convention: largest-magnitude component positive (the reference inherits LAPACK's, which is
    unspecified).  -> (sorted_eigenvalues (N,3), sorted_eigenvectors (N,3,3), quaternions (N,4)) float32."""
    c = _c(cov6, np.float64).reshape(-1, 6)
    n = c.shape[0]
    full = np.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], 1)
    ev, V = np.linalg.eigh(full)                                   # ascending; columns = eigenvectors
    m = np.abs(V).argmax(1)                                        # per column: row of the largest |component| (first on ties)
    sgn = np.sign(V[np.arange(n)[:, None], m, np.arange(3)[None, :]])
    sgn[sgn == 0] = 1
    V = V * sgn[:, None, :]
    corr = np.abs(V.transpose(0, 2, 1)).argmax(2)                  # (N,3): axis claimed by eigenvector k (gaussian_model.py:251-254)
    vals = np.zeros((n, 3), np.float32)
    vecs = np.zeros((n, 3, 3), np.float32)
    rows = np.arange(n)
    for k in range(3):                                             # scatter_ on the CPU: the later index wins
        vals[rows, corr[:, k]] = ev[:, k].astype(np.float32)
        vecs[rows, corr[:, k], :] = V[:, k, :].astype(np.float32)  # ROW k of the eigenvector matrix (gaussian_model.py:262)
    return vals, vecs, quaternions_reference(vecs)


def quaternions_reference(M):
    """matrices_to_quaternions (general_utils.py:94-100) in float32: w = sqrt(1 + trace) / 2, no branch."""
    M = np.asarray(M, np.float32)
    with np.errstate(all="ignore"):
        w = np.sqrt(np.float32(1) + (M[:, 0, 0] + M[:, 1, 1] + M[:, 2, 2])) / np.float32(2)
        d = np.float32(4) * w
        return np.stack([w, (M[:, 2, 1] - M[:, 1, 2]) / d, (M[:, 0, 2] - M[:, 2, 0]) / d, (M[:, 1, 0] - M[:, 0, 1]) / d], -1).astype(np.float32)


def cov_from_normals(normals, epsilon=1e-3):
    """Open3D GeneralizedICP.cpp, InitializePointCloudForGeneralizedICP + GetRotationFromE1ToX (numpy restatement;
    parity unpinned like the rest of the ICP half): C_i = Rx diag(eps, 1, 1) Rx^T, (n,3,3) float64."""
    nrm = np.asarray(normals, np.float64)
    n = nrm.shape[0]
    e1 = np.array([1.0, 0.0, 0.0])
    v = np.cross(np.broadcast_to(e1, nrm.shape), nrm)
    c = nrm @ e1
    sv = np.zeros((n, 3, 3))
    sv[:, 0, 1], sv[:, 0, 2] = -v[:, 2], v[:, 1]
    sv[:, 1, 0], sv[:, 1, 2] = v[:, 2], -v[:, 0]
    sv[:, 2, 0], sv[:, 2, 1] = -v[:, 1], v[:, 0]
    with np.errstate(all="ignore"):
        R = np.eye(3)[None] + sv + (sv @ sv) / (1.0 + c)[:, None, None]
    R[c < -0.99] = np.eye(3)
    C = np.diag([epsilon, 1.0, 1.0])
    return R @ C @ R.transpose(0, 2, 1)
