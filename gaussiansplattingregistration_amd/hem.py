"""Host side of the HEM half: one ``HemMixture`` is one ``hem::Mixture`` of the reference
(``src/cpp_ext/include/mixture.hpp:47-76``), living on one MI355X behind the C ABI.

Arrays may be numpy arrays (staged through host memory by the library) or PyTorch-ROCm tensors on
the context's device (zero-copy: only ``data_ptr()`` crosses the ABI).  PyTorch is plumbing here --
device memory and streams -- all arithmetic is in the HIP kernels of ``csrc/hem.hip``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

try:  # torch is optional for host-array use
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = ["HemMixture", "create_mixture"]


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _prep(a, shape, device_index):
    """-> (pointer, keepalive, on_device) for a float32 array of the given shape."""
    if _is_tensor(a):
        if a.is_cuda:
            if a.device.index != device_index:
                raise RuntimeError(f"tensor lives on {a.device}, context on cuda:{device_index}")
            t = a.detach().to(torch.float32).reshape(shape).contiguous()
            return t.data_ptr(), t, True
        a = a.detach().cpu().numpy()
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(shape))
    return arr.ctypes.data, arr, False


class HemMixture:
    """Hierarchical-EM mixture on the GPU.

    Parameters follow ``hem::Mixture`` / ``GaussianMixtureParams`` (``src/params/merge_parameters.py:5-10``):
    ``hem_reduction`` (rho), ``distance_delta``, ``color_delta``, ``decay_rate``.
    ``rng_mode``: ``"glibc"`` replays the reference's libc ``rand()`` stream (parity), ``"hash"`` is a
    counter-based device generator with the same distribution.
    """

    def __init__(self, hem_reduction=3.0, distance_delta=3.0, color_delta=2.5, decay_rate=1.0, device=0,
                 stream=None, rng_mode="glibc", rng_seed=1, rng_skip=0):
        self._L = _lib.load(require_device=True)
        self.device = int(device)
        if stream is None and torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _lib.check(self._L.gsr_hem_create(C.byref(h), self.device, C.c_void_p(stream or 0)), "gsr_hem_create")
        self._h = h
        self._borrowed = None
        self._out_cur = self._out_prev = None           # run_level(out=...): the caller's tensors the current / the previous level lives in
        _lib.check(self._L.gsr_hem_set_params(h, hem_reduction, distance_delta, color_delta, decay_rate), "gsr_hem_set_params")
        mode = {"glibc": _lib.GSR_RNG_GLIBC, "hash": _lib.GSR_RNG_HASH}[rng_mode]
        _lib.check(self._L.gsr_hem_set_rng(h, mode, rng_seed, rng_skip), "gsr_hem_set_rng")
        self.F = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsr_hem_destroy(self._h)
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

    def set_rng(self, mode="glibc", seed=1, skip=0):
        """Reposition the parent-flag stream (e.g. to replay a fresh reference process on a reused context)."""
        m = {"glibc": _lib.GSR_RNG_GLIBC, "hash": _lib.GSR_RNG_HASH}[mode]
        _lib.check(self._L.gsr_hem_set_rng(self._h, m, seed, skip), "gsr_hem_set_rng")

    def set_shard(self, rank, world, allreduce=None, allgather=None):
        """Work-sharded levels: this context evaluates the parents of slab ``rank`` of ``world``.
        ``allreduce(tensor)`` must sum a float32 CUDA tensor over the ranks in place (``torch.distributed.all_reduce``);
        ``allgather(send, recv)`` must fill the uint8 CUDA tensor ``recv`` (world x len(send)) with every rank's ``send``
        (``torch.distributed.all_gather_into_tensor``); see ``parallel.hem_sharded``."""
        if allreduce is None or allgather is None or world <= 1:
            self._shard_cb = None
            _lib.check(self._L.gsr_hem_set_shard(self._h, 0, 1, _lib.ALLREDUCE_DEV_FN(), _lib.ALLGATHER_DEV_FN(), None), "gsr_hem_set_shard")
            return
        dev = self.device

        class _Ptr:       # expose a raw device buffer to torch without a copy
            def __init__(self, ptr, count, typestr):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 3}

        def _reduce(ptr, count, _user):
            try:
                t = torch.as_tensor(_Ptr(int(ptr), int(count), "<f4"), device=torch.device("cuda", dev))
                allreduce(t)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def _gather(send, recv, nbytes, _user):
            try:
                s = torch.as_tensor(_Ptr(int(send), int(nbytes), "|u1"), device=torch.device("cuda", dev))
                r = torch.as_tensor(_Ptr(int(recv), int(nbytes) * int(world), "|u1"), device=torch.device("cuda", dev))
                allgather(s, r)
                torch.cuda.synchronize(dev)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        self._shard_cb = (_lib.ALLREDUCE_DEV_FN(_reduce), _lib.ALLGATHER_DEV_FN(_gather))
        _lib.check(self._L.gsr_hem_set_shard(self._h, int(rank), int(world), self._shard_cb[0], self._shard_cb[1], None), "gsr_hem_set_shard")

    # -- level 0 -----------------------------------------------------------------------------------
    def set_level0(self, xyz, colors, opacities, covariance, features, borrow=False):
        """``MixtureLevel.CreateMixtureLevel`` argument order (``mixturelevel.hpp:17-22``).
        ``borrow=True`` (device tensors only): the library reads the caller's tensors in place instead of copying them
        (1.19 GB at 5 M splats); they must not change until the next ``run_level`` has returned -- this object keeps
        references to them until then."""
        n = int(xyz.shape[0]) if hasattr(xyz, "shape") else len(xyz)
        if n == 0:
            F = 0
        else:
            f = features
            F = int(f.shape[1]) if hasattr(f, "shape") and len(f.shape) == 2 else (len(f[0]) if len(f) else 0)
        self.F = F
        px, kx, dx = _prep(xyz, (n, 3), self.device)
        pc, kc, dc = _prep(colors, (n, 3), self.device)
        pv, kv, dv = _prep(covariance, (n, 6), self.device)
        po, ko, do = _prep(opacities, (n,), self.device)
        if F > 0:
            ps, ks, ds = _prep(features, (n, F), self.device)
        else:
            ps, ks, ds = 0, None, dx
        if len({dx, dc, dv, do, ds}) != 1:
            raise RuntimeError("all level-0 arrays must live in the same place (all host or all device)")
        if dx and torch is not None:
            torch.cuda.current_stream(self.device).synchronize()
        mode = (2 if borrow else 1) if dx else 0
        _lib.check(self._L.gsr_hem_set_level0(self._h, px, pc, pv, po, ps, n, F, mode), "gsr_hem_set_level0")
        self._borrowed = (kx, kc, kv, ko, ks) if mode == 2 else None
        self._out_cur = self._out_prev = None
        del kx, kc, kv, ko, ks

    # -- spatially partitioned levels (one large cloud over several GPUs) ------------------------------------------------
    def set_comm(self, comm):
        """The communicator (``comm.Comm``) the partitioned levels exchange through; ``None`` = single GPU."""
        self._comm = comm
        _lib.check(self._L.gsr_hem_set_comm(self._h, comm.handle if comm is not None else None), "gsr_hem_set_comm")

    def set_level0_part(self, xyz, colors, opacities, covariance, features, gid, n_global):
        """This rank's OWNED components of a cloud of ``n_global`` with their global indices ``gid`` (ascending, uint32)."""
        n = int(xyz.shape[0])
        f = features
        F = int(f.shape[1]) if hasattr(f, "shape") and len(f.shape) == 2 else 0
        self.F = F
        px, kx, dx = _prep(xyz, (n, 3), self.device)
        pc, kc, dc = _prep(colors, (n, 3), self.device)
        pv, kv, dv = _prep(covariance, (n, 6), self.device)
        po, ko, do = _prep(opacities, (n,), self.device)
        ps, ks, ds = _prep(features, (n, F), self.device) if F > 0 else (0, None, dx)
        if len({dx, dc, dv, do, ds}) != 1:
            raise RuntimeError("all level-0 arrays must live in the same place (all host or all device)")
        n_gid = int(gid.numel()) if _is_tensor(gid) else int(np.asarray(gid).size)      # (a plain list of indices is fine too)
        if n_gid != n:
            raise RuntimeError(f"gid has {n_gid} entries for {n} components")
        if dx:
            g = (gid.detach() if _is_tensor(gid) else torch.as_tensor(np.asarray(gid, np.int64))).to(xyz.device, torch.int32).contiguous()
            pg, kg = g.data_ptr(), g
            torch.cuda.current_stream(self.device).synchronize()
        else:
            kg = np.ascontiguousarray(gid.cpu().numpy() if _is_tensor(gid) else np.asarray(gid), dtype=np.uint32)
            pg = kg.ctypes.data
        _lib.check(self._L.gsr_hem_set_level0_part(self._h, px, pc, pv, po, ps, pg, n, int(n_global), F, 1 if dx else 0), "gsr_hem_set_level0_part")
        self._out_cur = self._out_prev = None

    def gids(self):
        """Global indices (uint32 numpy) of the current partitioned level's owned rows."""
        out = np.empty(self.size, np.uint32)
        _lib.check(self._L.gsr_hem_get_gids(self._h, out.ctypes.data, 0), "gsr_hem_get_gids")
        return out

    def part_stats(self):
        s = (C.c_int64 * 8)()
        _lib.check(self._L.gsr_hem_get_part_stats(self._h, s), "gsr_hem_get_part_stats")
        ms = (C.c_float * 4)()
        _lib.check(self._L.gsr_hem_get_part_ms(self._h, ms), "gsr_hem_get_part_ms")
        return {"ghosts": s[0], "rows_sent": s[1], "halo_bytes_received": s[2], "sum_exchange_bytes_received": s[3], "parents_global": s[4],
                "orphans_global": s[5], "dropped_global": s[6], "n_global": s[7], "ms_halo_rows": float(ms[0]), "ms_halo_sh_overlapped": float(ms[1])}

    def set_state(self, parent_mask=None, weight=None):
        pm = None if parent_mask is None else np.ascontiguousarray(parent_mask, dtype=np.uint8)
        w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float32)
        _lib.check(self._L.gsr_hem_set_state(self._h, None if pm is None else pm.ctypes.data,
                                             None if w is None else w.ctypes.data), "gsr_hem_set_state")

    # -- levels ------------------------------------------------------------------------------------
    def run_level(self, out=None):
        """One ``createClusterLevel``.  Returns ``(n_out, n_dropped)``.

        ``out``: a dict of CUDA float32 tensors ``xyz (R,3) color (R,3) cov6 (R,6) opacity (R,) sh (R,F)`` with R >= the size of the level
        being reduced -- the new level is written straight into them (``gsr_hem_set_output``) and they then ARE the current level: no
        copy on the way out, none on the way into the next level; ``get_level(as_torch=True)`` afterwards returns views of their first
        ``n_out`` rows.  They must stay untouched until the ``run_level`` after this one has returned (this object keeps them alive)."""
        n_out, n_drop = C.c_int64(0), C.c_int64(0)
        if out is not None:
            rows = int(out["xyz"].shape[0])
            for k, w in (("xyz", 3), ("color", 3), ("cov6", 6), ("opacity", 1), ("sh", self.F)):
                t = out.get(k)
                if w == 0 and k == "sh":
                    continue
                if t is None or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != rows * w or t.device.index != self.device:
                    raise RuntimeError(f"run_level(out=...): '{k}' must be a contiguous float32 CUDA tensor of {rows} x {w} on cuda:{self.device}")
            torch.cuda.current_stream(self.device).synchronize()
            sh = out["sh"].data_ptr() if self.F > 0 else None
            _lib.check(self._L.gsr_hem_set_output(self._h, out["xyz"].data_ptr(), out["color"].data_ptr(), out["cov6"].data_ptr(), out["opacity"].data_ptr(),
                                                  sh, rows), "gsr_hem_set_output")
        _lib.check(self._L.gsr_hem_run_level(self._h, C.byref(n_out), C.byref(n_drop)), "gsr_hem_run_level")
        self._borrowed = None                       # a borrowed level 0 has been consumed
        self._out_prev = getattr(self, "_out_cur", None)         # (the level just consumed: free to go once this call has returned)
        self._out_cur = out
        return int(n_out.value), int(n_drop.value)

    def new_arena(self, rows, normals=False):
        """Uninitialised arenas for ``run_levels``: ``rows`` rows of every exported array (+ float64 normals)."""
        dev = torch.device("cuda", self.device)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        rows = int(rows)
        a = {"xyz": e(rows, 3), "color": e(rows, 3), "cov6": e(rows, 6), "opacity": e(rows), "sh": e(rows, self.F)}
        if normals:
            a["normals"] = torch.empty((rows, 3), dtype=torch.float64, device=dev)
        return a

    def run_levels(self, n_levels, arena=None, normals0=None):
        """``MixtureCreator.CreateMixture(clusterLevel, ...)`` in ONE library call (``gsr_hem_run_levels``; mixture_wrapper.cpp:10-18):
        ``n_levels`` clustering levels on the current level, written one behind the other into the CUDA tensors of ``arena``
        (``new_arena``; default: ``n_levels`` x the current level's rows, always enough -- a level never grows) -- no return to Python between
        the levels.  Returns ``(levels, stats)``: per level a dict of VIEWS ``xyz color cov6 opacity sh`` (+ ``normals`` when the arena has
        a ``normals`` array: the normals leave with the level) and the dict ``stats()`` would have returned after it (+ ``dropped_now``).
        ``normals0``: a (n, 3) float64 CUDA tensor that receives the normals of the level the call starts from.
        The arena must stay untouched until the next ``run_level(s)`` / ``set_level0`` has returned (this object keeps it alive)."""
        n_levels = int(n_levels)
        n0 = self.size
        if arena is None:                 # always enough: a level never grows (the isotropic bench cloud needs 1.5 x, a cloud of discs and needles 1.7 x for three levels)
            arena = self.new_arena(max(64, n_levels * (n0 + 64)))
        rows = int(arena["xyz"].shape[0])
        for k, w in (("xyz", 3), ("color", 3), ("cov6", 6), ("opacity", 1), ("sh", self.F)):
            t = arena.get(k)
            if w == 0 and k == "sh":
                continue
            if t is None or not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != rows * w or t.device.index != self.device:
                raise RuntimeError(f"run_levels(arena=...): '{k}' must be a contiguous float32 CUDA tensor of {rows} x {w} on cuda:{self.device}")
        nrm = arena.get("normals")
        for t, r in ((nrm, rows), (normals0, n0)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous() or t.numel() != r * 3 or t.device.index != self.device):
                raise RuntimeError(f"run_levels: normals must be contiguous float64 CUDA tensors of rows x 3 on cuda:{self.device}")
        reports = (_lib.HemLevelReport * max(1, n_levels))()
        torch.cuda.current_stream(self.device).synchronize()
        rc = self._L.gsr_hem_run_levels(self._h, n_levels, arena["xyz"].data_ptr(), arena["color"].data_ptr(), arena["cov6"].data_ptr(),
                                        arena["opacity"].data_ptr(), arena["sh"].data_ptr() if self.F > 0 else None,
                                        nrm.data_ptr() if nrm is not None else None, normals0.data_ptr() if normals0 is not None else None,
                                        rows, C.cast(reports, C.c_void_p))
        self._borrowed = None
        self._arena = arena                         # keeps the current level's memory alive
        self._out_prev, self._out_cur = None, None
        _lib.check(rc, "gsr_hem_run_levels")
        levels, stats = [], []
        for k in range(n_levels):
            R = reports[k]
            o, n = int(R.offset_rows), int(R.rows)
            lv = {f: arena[f][o:o + n] for f in ("xyz", "color", "cov6", "opacity", "sh")}
            if nrm is not None:
                lv["normals"] = nrm[o:o + n]
            levels.append(lv)
            stats.append(self._stats_dict(R.stats, R.phase_ms, R.stats_ex, R.kernel_ms, int(R.rng_position), dropped_now=int(R.dropped)))
        if n_levels > 0:
            last = levels[-1]
            self._out_cur = {f: last[f] for f in ("xyz", "color", "cov6", "opacity", "sh")}      # get_level(as_torch=True) returns these views
        return levels, stats

    def new_output(self, rows=None):
        """Uninitialised output arrays for ``run_level(out=...)``: ``rows`` defaults to the current level's size (always enough)."""
        rows = self.size if rows is None else int(rows)
        dev = torch.device("cuda", self.device)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        return {"xyz": e(rows, 3), "color": e(rows, 3), "cov6": e(rows, 6), "opacity": e(rows), "sh": e(rows, self.F)}

    @property
    def size(self):
        n, F = C.c_int64(0), C.c_int32(0)
        _lib.check(self._L.gsr_hem_level_size(self._h, C.byref(n), C.byref(F)), "gsr_hem_level_size")
        return int(n.value)

    def get_level(self, as_torch=False, with_state=False):
        """Current level as a dict of arrays: xyz, color, cov6, opacity, sh (+ weight, is_parent)."""
        n, F = self.size, self.F
        cur = getattr(self, "_out_cur", None)
        if as_torch and cur is not None:            # the level already lives in the caller's tensors (run_level(out=...)): views, no copy
            out = {k: cur[k][:n] for k in ("xyz", "color", "cov6", "opacity")}
            # (F == 0: run_level(out=...) lets the caller leave 'sh' out)
            out["sh"] = cur["sh"][:n] if (F > 0 or cur.get("sh") is not None) else torch.empty((n, 0), dtype=torch.float32, device=cur["xyz"].device)
            if with_state:
                dev = torch.device("cuda", self.device)
                out["weight"] = torch.empty((n,), dtype=torch.float32, device=dev)
                out["is_parent"] = torch.empty((n,), dtype=torch.uint8, device=dev)
                _lib.check(self._L.gsr_hem_get_level(self._h, None, None, None, None, None, out["weight"].data_ptr(), out["is_parent"].data_ptr(), 1), "gsr_hem_get_level")
            return out
        if as_torch:
            dev = torch.device("cuda", self.device)
            out = {"xyz": torch.empty((n, 3), dtype=torch.float32, device=dev),
                   "color": torch.empty((n, 3), dtype=torch.float32, device=dev),
                   "cov6": torch.empty((n, 6), dtype=torch.float32, device=dev),
                   "opacity": torch.empty((n,), dtype=torch.float32, device=dev),
                   "sh": torch.empty((n, F), dtype=torch.float32, device=dev)}
            if with_state:
                out["weight"] = torch.empty((n,), dtype=torch.float32, device=dev)
                out["is_parent"] = torch.empty((n,), dtype=torch.uint8, device=dev)
            ptr = lambda k: out[k].data_ptr() if k in out else None
            torch.cuda.current_stream(self.device).synchronize()
        else:
            out = {"xyz": np.empty((n, 3), np.float32), "color": np.empty((n, 3), np.float32),
                   "cov6": np.empty((n, 6), np.float32), "opacity": np.empty((n,), np.float32),
                   "sh": np.empty((n, F), np.float32)}
            if with_state:
                out["weight"] = np.empty((n,), np.float32)
                out["is_parent"] = np.empty((n,), np.uint8)
            ptr = lambda k: out[k].ctypes.data if k in out else None
        _lib.check(self._L.gsr_hem_get_level(self._h, ptr("xyz"), ptr("color"), ptr("cov6"), ptr("opacity"),
                                             ptr("sh") if F > 0 else None, ptr("weight"), ptr("is_parent"),
                                             1 if as_torch else 0), "gsr_hem_get_level")
        return out

    def set_timing(self, level: int):
        """How much a level times: 0 nothing, 1 (the default) the level and the launches of k_select / k_mstep, 2 every phase and the
        partition kernels too (``stats()``'s other ``ms_*`` entries read 0 below 2).  Events between kernels are not free: 22 per level
        at 2, 0.1 ms of a 5 M-splat level (gsr_hem_set_timing)."""
        _lib.check(self._L.gsr_hem_set_timing(self._h, int(level)), "gsr_hem_set_timing")

    def stats(self):
        s = (C.c_int64 * 8)()
        _lib.check(self._L.gsr_hem_get_stats(self._h, s), "gsr_hem_get_stats")
        t = (C.c_float * 8)()
        _lib.check(self._L.gsr_hem_get_phase_ms(self._h, t), "gsr_hem_get_phase_ms")
        pos = C.c_uint64(0)
        _lib.check(self._L.gsr_hem_get_rng_position(self._h, C.byref(pos)), "gsr_hem_get_rng_position")
        x = (C.c_int64 * 8)()
        _lib.check(self._L.gsr_hem_get_stats_ex(self._h, x), "gsr_hem_get_stats_ex")
        km = (C.c_float * 8)()
        _lib.check(self._L.gsr_hem_get_kernel_ms(self._h, km), "gsr_hem_get_kernel_ms")
        return self._stats_dict(s, t, x, km, int(pos.value))

    @staticmethod
    def _stats_dict(s, t, x, km, rng_draws, **extra):
        # round_trips: host round trips of the level (1 = asynchronous: one answer behind the level's last kernel; a synchronous level takes
        # four to six); schedule: 0 synchronous, 1 asynchronous, 2 an asynchronous attempt whose buffers were too small, rerun synchronously
        return {**extra, "irregular": x[0], "one_pass": x[1], "partition_overflow": x[2], "heavy_parents": x[3], "heavy_work_items": x[4], "max_pairs_of_a_parent": x[5],
                "round_trips": x[6], "schedule": x[7],
                "ms_k_select": km[0], "ms_k_mstep": km[1], "ms_k_partition": km[2], "ms_k_bucket_sum": km[3],
                "parents": s[0], "pairs": s[1], "orphans": s[2], "dropped": s[3], "candidates": s[4], "cells": s[5],
                "n_in": s[6], "n_out": s[7], "ms_grid": t[0], "ms_select": t[1], "ms_sumlw": t[2], "ms_mstep": t[3],
                "ms_flags": t[4], "ms_level": t[5], "ms_k_select_count": t[6], "ms_k_select_fill": t[7],
                "rng_draws": rng_draws}


def create_mixture(cloud: dict, cluster_level: int, hem_reduction=3.0, distance_delta=3.0, color_delta=2.5,
                   decay_rate=1.0, device=0, as_torch=False, rng_mode="glibc", rng_seed=1, rng_skip=0, with_state=False):
    """``MixtureCreator.CreateMixture`` on a dict cloud: returns (levels, stats), level 0 dropped.  ``with_state``: the levels also
    carry ``weight`` and ``is_parent`` (internal state the reference never exports)."""
    with HemMixture(hem_reduction, distance_delta, color_delta, decay_rate, device=device, rng_mode=rng_mode,
                    rng_seed=rng_seed, rng_skip=rng_skip) as m:
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        levels, stats = [], []
        for _ in range(int(cluster_level)):
            m.run_level()
            stats.append(m.stats())
            levels.append(m.get_level(as_torch=as_torch, with_state=with_state))
        return levels, stats
