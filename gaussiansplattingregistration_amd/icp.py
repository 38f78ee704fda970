"""Host side of the ICP half: ``registration_icp`` semantics of Open3D 0.16.0 (what the reference's
``do_icp_registration`` reaches, ``src/utils/local_registration_util.py:76-100``) on one MI355X, with
an optional ``torch.distributed`` all-reduce of the accumulator vector for multi-GPU source splits.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = ["IcpContext", "registration_icp_arrays", "normals_from_cov", "normals_knn", "cov_from_normals", "RegistrationResult"]

KIND_POINT_TO_POINT = 0
KIND_POINT_TO_PLANE = 1
KIND_GENERALIZED = 2
KIND_COLORED = 3
LOSS_L2, LOSS_TUKEY, LOSS_CAUCHY, LOSS_GM, LOSS_HUBER = 0, 1, 2, 3, 4


class RegistrationResult:
    """The three fields the reference's callers read (``registration_controller.py:145-148``)."""

    def __init__(self, transformation, fitness, inlier_rmse, iterations=0, correspondence_set=None):
        self.transformation = transformation
        self.fitness = fitness
        self.inlier_rmse = inlier_rmse
        self.iterations = iterations
        self.correspondence_set = correspondence_set

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and iterations={self.iterations}")


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _prep(a, shape, dtype, device_index):
    tdt = {np.float32: "float32", np.float64: "float64"}[dtype]
    if _is_tensor(a):
        if a.is_cuda:
            if a.device.index != device_index:
                raise RuntimeError(f"tensor lives on {a.device}, context on cuda:{device_index}")
            t = a.detach().to(getattr(torch, tdt)).reshape(shape).contiguous()
            return t.data_ptr(), t, True
        a = a.detach().cpu().numpy()
    arr = np.ascontiguousarray(np.asarray(a, dtype=dtype).reshape(shape))
    return arr.ctypes.data, arr, False


class IcpContext:
    """Target index + source buffer on one GPU."""

    def __init__(self, device=0, stream=None):
        self._L = _lib.load(require_device=True)
        self.device = int(device)
        if stream is None and torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _lib.check(self._L.gsr_icp_create(C.byref(h), self.device, C.c_void_p(stream or 0)), "gsr_icp_create")
        self._h = h
        self._cb = None
        self._stream = int(stream or 0)
        self._comm = None
        self.n_source = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsr_icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _sync_torch(self):
        if torch is not None and torch.cuda.is_available():
            torch.cuda.current_stream(self.device).synchronize()

    def set_target(self, xyz, normals, max_corr):
        n = int(xyz.shape[0])
        px, kx, dx = _prep(xyz, (n, 3), np.float32, self.device)
        pn, kn, dn = (None, None, dx) if normals is None else _prep(normals, (n, 3), np.float64, self.device)
        if dn != dx:
            raise RuntimeError("target points and normals must live in the same place (both host or both device)")
        if dx:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_target(self._h, px, pn, n, float(max_corr), 1 if dx else 0), "gsr_icp_set_target")
        self.n_target = n

    def set_source(self, xyz):
        n = int(xyz.shape[0])
        if n == 0:                         # an empty shard of a multi-GPU source split
            _lib.check(self._L.gsr_icp_set_source(self._h, None, 0, 0), "gsr_icp_set_source")
            self.n_source = 0
            return
        px, kx, dx = _prep(xyz, (n, 3), np.float32, self.device)
        if dx:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_source(self._h, px, n, 1 if dx else 0), "gsr_icp_set_source")
        self.n_source = n

    @staticmethod
    def _cov6(cov):
        """(N,6) [xx,xy,xz,yy,yz,zz] or (N,3,3) covariances -> contiguous (N,6) float64 (numpy or tensor kept as is)."""
        if _is_tensor(cov):
            import torch
            if cov.dim() == 3:
                cov = torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
            return cov.to(torch.float64).contiguous()
        cov = np.asarray(cov)
        if cov.ndim == 3:
            cov = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], 1)
        return np.ascontiguousarray(cov, dtype=np.float64)

    def set_target_cov(self, cov):
        """Target covariances for generalized ICP, in the order of the points given to ``set_target``."""
        c6 = self._cov6(cov)
        p, keep, on_dev = _prep(c6, (self.n_target, 6), np.float64, self.device)
        if on_dev:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_target_cov(self._h, p, 1 if on_dev else 0), "gsr_icp_set_target_cov")

    def set_source_cov(self, cov):
        """Source covariances (original frame; the library rotates them with the current transform)."""
        c6 = self._cov6(cov)
        p, keep, on_dev = _prep(c6, (self.n_source, 6), np.float64, self.device)
        if on_dev:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_source_cov(self._h, p, 1 if on_dev else 0), "gsr_icp_set_source_cov")

    def set_target_color(self, rgb):
        """Target colours (N,3) for colored ICP; also prepares the colour gradients (needs target normals)."""
        p, keep, on_dev = _prep(rgb, (self.n_target, 3), np.float64, self.device)
        if on_dev:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_target_color(self._h, p, 1 if on_dev else 0), "gsr_icp_set_target_color")

    def set_source_color(self, rgb):
        p, keep, on_dev = _prep(rgb, (self.n_source, 3), np.float64, self.device)
        if on_dev:
            self._sync_torch()
        _lib.check(self._L.gsr_icp_set_source_color(self._h, p, 1 if on_dev else 0), "gsr_icp_set_source_color")

    def set_lambda_geometric(self, lambda_geometric):
        _lib.check(self._L.gsr_icp_set_lambda_geometric(self._h, float(lambda_geometric)), "gsr_icp_set_lambda_geometric")

    def color_gradient(self):
        """The target's colour gradients (N,3) float64 in the caller's point order (test / inspection hook)."""
        out = np.empty((self.n_target, 3), np.float64)
        _lib.check(self._L.gsr_icp_get_color_gradient(self._h, out.ctypes.data), "gsr_icp_get_color_gradient")
        return out

    def set_allreduce(self, fn, n_source_global):
        """``fn(numpy float64[32]) -> None`` must sum the vector over all ranks in place."""
        if fn is None:
            self._cb = None
            _lib.check(self._L.gsr_icp_set_allreduce(self._h, _lib.ALLREDUCE_FN(), None, 0), "gsr_icp_set_allreduce")
            return

        def _tramp(buf, length, _user):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(length,))
                fn(arr)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        self._cb = _lib.ALLREDUCE_FN(_tramp)
        _lib.check(self._L.gsr_icp_set_allreduce(self._h, self._cb, None, int(n_source_global)), "gsr_icp_set_allreduce")

    def set_allreduce_device(self, fn, n_source_global):
        """``fn(tensor float64[32] on this device) -> None`` must sum the vector over all ranks in place, ordered on the
        current stream (``torch.distributed.all_reduce`` is).  The iteration loop stays device resident; every rank
        solves from the identical reduced vector.  ``fn = None`` restores the single-GPU loop."""
        if fn is None:
            self._cb_dev = None
            _lib.check(self._L.gsr_icp_set_allreduce_dev(self._h, _lib.ALLREDUCE_DEV_FN(), None, 0), "gsr_icp_set_allreduce_dev")
            return
        dev = self.device

        class _Ptr:
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3}

        ext = torch.cuda.ExternalStream(self._stream, device=torch.device("cuda", dev)) if self._stream else None

        def _tramp(ptr, count, _user):
            try:
                t = torch.as_tensor(_Ptr(int(ptr), int(count)), device=torch.device("cuda", dev))
                # the collective must be ordered on the CONTEXT's stream (the library enqueues the reduction in front of it and
                # the solve behind it there), whatever torch's current stream is when the library calls back
                with torch.cuda.stream(ext if ext is not None else torch.cuda.default_stream(dev)):
                    fn(t)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        self._cb_dev = _lib.ALLREDUCE_DEV_FN(_tramp)
        _lib.check(self._L.gsr_icp_set_allreduce_dev(self._h, self._cb_dev, None, int(n_source_global)), "gsr_icp_set_allreduce_dev")

    def set_comm(self, comm, n_source_global=0):
        """Multi-GPU source split through a library communicator (``comm.Comm``): per iteration ONE all-reduce of 32 float64,
        enqueued by the library itself on this context's stream (RCCL transport).  ``comm = None`` restores single-GPU."""
        self._comm = comm                      # keep it alive while the context may use it
        _lib.check(self._L.gsr_icp_set_comm(self._h, comm.handle if comm is not None else None, int(n_source_global)), "gsr_icp_set_comm")

    def accumulate(self, T, kind=0, loss=0, k=0.0):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
        acc = np.zeros(_lib.GSR_ICP_ACC_LEN, np.float64)
        _lib.check(self._L.gsr_icp_accumulate(self._h, T.ctypes.data, kind, loss, float(k), acc.ctypes.data), "gsr_icp_accumulate")
        return acc

    def register(self, init=None, kind=0, loss=0, k=0.0, rel_fitness=1e-6, rel_rmse=1e-6, max_iter=30):
        init = np.eye(4) if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
        T = np.empty((4, 4), np.float64)
        fit, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
        _lib.check(self._L.gsr_icp_register(self._h, init.ctypes.data, kind, loss, float(k), float(rel_fitness), float(rel_rmse),
                                            int(max_iter), T.ctypes.data, C.byref(fit), C.byref(rmse), C.byref(it)),
                   "gsr_icp_register")
        return {"transformation": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iterations": int(it.value)}

    def register_clouds(self, src_xyz, tgt_xyz, tgt_normals, max_corr, init=None, kind=0, loss=0, k=0.0, rel_fitness=1e-6, rel_rmse=1e-6, max_iter=30):
        """``registration_icp(source, target, max_correspondence_distance, init, estimation, criteria)`` in ONE library call
        (``gsr_icp_register_clouds``): target index, source order and the iteration loop without a return to Python or a stream
        synchronisation between them.  Point-to-point / point-to-plane, one process; the same result as ``set_target`` + ``set_source`` +
        ``register``.  All arrays on the host or all on the device."""
        ns, nt = int(src_xyz.shape[0]), int(tgt_xyz.shape[0])
        ps, ks, ds = _prep(src_xyz, (ns, 3), np.float32, self.device)
        pt, kt, dt = _prep(tgt_xyz, (nt, 3), np.float32, self.device)
        pn, kn, dn = (None, None, dt) if tgt_normals is None else _prep(tgt_normals, (nt, 3), np.float64, self.device)
        if not (ds == dt == dn):
            raise RuntimeError("register_clouds: source, target and normals must live in the same place (all host or all device)")
        if dt:
            self._sync_torch()
        init = np.eye(4) if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
        T = np.empty((4, 4), np.float64)
        fit, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
        _lib.check(self._L.gsr_icp_register_clouds(self._h, ps, ns, pt, pn, nt, 1 if dt else 0, float(max_corr), init.ctypes.data, kind, loss, float(k),
                                                   float(rel_fitness), float(rel_rmse), int(max_iter), T.ctypes.data, C.byref(fit), C.byref(rmse),
                                                   C.byref(it)), "gsr_icp_register_clouds")
        self.n_target, self.n_source = nt, ns
        return {"transformation": T, "fitness": fit.value, "inlier_rmse": rmse.value, "iterations": int(it.value)}

    def register_multiscale(self, entries, init=None, kind=0, loss=0, k=0.0, rel_fitness=1e-6, rel_rmse=1e-6):
        """The coarse-to-fine schedule in ONE library call (``gsr_icp_register_multiscale``; qt_multiscale_registrator.py:197-236).  ``entries``: coarsest
        first, each ``(src_xyz, tgt_xyz, tgt_normals or None, max_corr, max_iter)`` with all arrays on the host or all on the device.  Returns the list of
        per-entry dicts (``init``, ``transformation``, ``fitness``, ``inlier_rmse``, ``iterations``, ``evaluations``, ``ms_build``, ``ms_iters``)."""
        n = len(entries)
        ent = (_lib.IcpEntry * max(1, n))()
        res = (_lib.IcpEntryResult * max(1, n))()
        keep, place = [], set()
        for i, (sx, tx, tn, mc, it) in enumerate(entries):
            ns, nt = int(sx.shape[0]), int(tx.shape[0])
            ps, ks, ds = _prep(sx, (ns, 3), np.float32, self.device)
            pt, kt, dt = _prep(tx, (nt, 3), np.float32, self.device)
            pn, kn, dn = (None, None, dt) if tn is None else _prep(tn, (nt, 3), np.float64, self.device)
            keep += [ks, kt, kn]
            place |= {ds, dt, dn}
            ent[i].src_xyz, ent[i].ns, ent[i].tgt_xyz, ent[i].tgt_normals, ent[i].nt = ps, ns, pt, pn, nt
            ent[i].max_corr, ent[i].max_iter = float(mc), int(it)
        if len(place) > 1:
            raise RuntimeError("register_multiscale: every array of every entry must live in the same place (all host or all device)")
        on_dev = bool(place.pop()) if place else False
        if on_dev:
            self._sync_torch()
        init = np.eye(4) if init is None else np.ascontiguousarray(init, dtype=np.float64).reshape(4, 4)
        T = np.empty((4, 4), np.float64)
        _lib.check(self._L.gsr_icp_register_multiscale(self._h, n, C.cast(ent, C.c_void_p), 1 if on_dev else 0, init.ctypes.data, kind, loss, float(k),
                                                       float(rel_fitness), float(rel_rmse), C.cast(res, C.c_void_p), T.ctypes.data), "gsr_icp_register_multiscale")
        out = []
        for i in range(n):
            r = res[i]
            out.append({"init": np.array(r.init_T[:], np.float64).reshape(4, 4), "transformation": np.array(r.T[:], np.float64).reshape(4, 4),
                        "fitness": r.fitness, "inlier_rmse": r.inlier_rmse, "iterations": int(r.iterations), "evaluations": int(r.evaluations),
                        "ms_build": float(r.ms_build), "ms_iters": float(r.ms_iters)})
        if n:
            self.n_target, self.n_source = int(entries[-1][1].shape[0]), int(entries[-1][0].shape[0])
        return out

    def correspondences(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
        idx = np.empty(self.n_source, np.int64)
        d2 = np.empty(self.n_source, np.float64)
        _lib.check(self._L.gsr_icp_correspondences(self._h, T.ctypes.data, idx.ctypes.data, d2.ctypes.data), "gsr_icp_correspondences")
        return idx, d2

    def timing(self):
        t = (C.c_float * 3)()
        _lib.check(self._L.gsr_icp_get_timing(self._h, t), "gsr_icp_get_timing")
        return {"ms_build": t[0], "ms_iters": t[1], "iter_kernels": int(t[2])}


def registration_icp_arrays(src_xyz, tgt_xyz, tgt_normals, init, kind=0, loss=0, k=0.0, max_corr=1.0, rel_fitness=1e-6,
                            rel_rmse=1e-6, max_iter=30, device=0, src_cov=None, tgt_cov=None, src_color=None, tgt_color=None,
                            lambda_geometric=None):
    """One ``registration_icp`` on raw arrays; returns dict(transformation, fitness, inlier_rmse, iterations).
    kind 2 (generalized ICP) needs ``src_cov`` / ``tgt_cov`` ((N,6) or (N,3,3)); kind 3 (colored ICP) needs target normals
    and ``src_color`` / ``tgt_color`` (N,3)."""
    with IcpContext(device=device) as c:
        c.set_target(tgt_xyz, tgt_normals, max_corr)
        c.set_source(src_xyz)
        if kind == 2:
            if src_cov is None or tgt_cov is None:
                raise RuntimeError("TransformationEstimationForGeneralizedICP requires source and target covariances")
            c.set_target_cov(tgt_cov)
            c.set_source_cov(src_cov)
        if kind == 3:
            if src_color is None or tgt_color is None:
                raise RuntimeError("ColoredICP requires the colours of both clouds")
            if lambda_geometric is not None:
                c.set_lambda_geometric(lambda_geometric)
            c.set_target_color(tgt_color)
            c.set_source_color(src_color)
        out = c.register(init, kind, loss, k, rel_fitness, rel_rmse, max_iter)
        out.update(c.timing())
        return out


def normals_from_cov(cov6, device=0):
    """float64 unit normals (n,3) = smallest-eigenvalue eigenvector of each splat covariance."""
    L = _lib.load(require_device=True)
    n = int(cov6.shape[0])
    if _is_tensor(cov6) and cov6.is_cuda:
        t = cov6.detach().to(torch.float32).reshape(n, 6).contiguous()
        out = torch.empty((n, 3), dtype=torch.float64, device=t.device)
        torch.cuda.current_stream(t.device.index).synchronize()
        _lib.check(L.gsr_normals_from_cov(t.data_ptr(), n, out.data_ptr(), 1, t.device.index,
                                          C.c_void_p(torch.cuda.current_stream(t.device.index).cuda_stream)), "gsr_normals_from_cov")
        return out
    a = np.ascontiguousarray(cov6.detach().cpu().numpy() if _is_tensor(cov6) else cov6, dtype=np.float32).reshape(n, 6)
    out = np.empty((n, 3), np.float64)
    _lib.check(L.gsr_normals_from_cov(a.ctypes.data, n, out.ctypes.data, 0, int(device), None), "gsr_normals_from_cov")
    return out


def normals_knn(xyz, knn=30, device=0):
    """float64 unit normals (n,3) of a cloud without covariances: Open3D ``estimate_normals()`` with its default
    ``KDTreeSearchParamKNN(30)`` (what the reference does to a sparse input cloud, ``point_cloud_converter.py:9-28``)."""
    L = _lib.load(require_device=True)
    n = int(xyz.shape[0])
    if _is_tensor(xyz) and xyz.is_cuda:
        t = xyz.detach().to(torch.float32).reshape(n, 3).contiguous()
        out = torch.empty((n, 3), dtype=torch.float64, device=t.device)
        torch.cuda.current_stream(t.device.index).synchronize()
        _lib.check(L.gsr_normals_knn(t.data_ptr(), n, int(knn), out.data_ptr(), 1, t.device.index,
                                     C.c_void_p(torch.cuda.current_stream(t.device.index).cuda_stream)), "gsr_normals_knn")
        return out
    a = np.ascontiguousarray(xyz.detach().cpu().numpy() if _is_tensor(xyz) else xyz, dtype=np.float32).reshape(n, 3)
    out = np.empty((n, 3), np.float64)
    _lib.check(L.gsr_normals_knn(a.ctypes.data, n, int(knn), out.ctypes.data, 0, int(device), None), "gsr_normals_knn")
    return out


def cov_from_normals(normals, epsilon=1e-3, device=0):
    """float64 covariances (n,6) [xx,xy,xz,yy,yz,zz] of Open3D's ``InitializePointCloudForGeneralizedICP``: a disc of
    thickness ``epsilon`` perpendicular to each normal -- what generalized ICP uses for a cloud without covariances."""
    L = _lib.load(require_device=True)
    n = int(normals.shape[0])
    if _is_tensor(normals) and normals.is_cuda:
        t = normals.detach().to(torch.float64).reshape(n, 3).contiguous()
        out = torch.empty((n, 6), dtype=torch.float64, device=t.device)
        torch.cuda.current_stream(t.device.index).synchronize()
        _lib.check(L.gsr_cov_from_normals(t.data_ptr(), n, float(epsilon), out.data_ptr(), 1, t.device.index,
                                          C.c_void_p(torch.cuda.current_stream(t.device.index).cuda_stream)), "gsr_cov_from_normals")
        return out
    a = np.ascontiguousarray(normals.detach().cpu().numpy() if _is_tensor(normals) else normals, dtype=np.float64).reshape(n, 3)
    out = np.empty((n, 6), np.float64)
    _lib.check(L.gsr_cov_from_normals(a.ctypes.data, n, float(epsilon), out.ctypes.data, 0, int(device), None), "gsr_cov_from_normals")
    return out
