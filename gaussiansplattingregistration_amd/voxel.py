"""Voxel down-sampling on the GPU: ctypes front of ``gsr_voxel_*`` (``include/gsr_hip.h``, ``csrc/voxel.hip``).

Reference behaviour: ``pc.voxel_down_sample(radius)`` of Open3D 0.16.0 at
``src/gui/workers/registration/qt_multiscale_registrator.py:127-128``.  Voxels come out in ascending (ix, iy, iz)
order (Open3D: hash-map order, implementation defined).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = ["voxel_down_sample"]


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


def _prep32(a, cols, device):
    """-> (pointer, keep-alive, on_device, n)"""
    if a is None:
        return None, None, None, 0
    if _is_tensor(a) and a.is_cuda:
        if a.device.index != device:
            raise RuntimeError(f"tensor lives on {a.device}, requested cuda:{device}")
        t = a.detach().to(torch.float32).reshape(-1, cols).contiguous()
        return t.data_ptr(), t, True, int(t.shape[0])
    if _is_tensor(a):
        a = a.detach().cpu().numpy()
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, cols))
    return arr.ctypes.data, arr, False, int(arr.shape[0])


def voxel_down_sample(xyz, voxel_size, cov6=None, color=None, device=0, as_torch=False):
    """Means per occupied voxel: ``(xyz (V,3), cov6 (V,6) | None, color (V,3) | None)`` in float64 (numpy, or cuda
    tensors with ``as_torch``).  Inputs: float32 arrays or tensors; all on the host or all on the device."""
    if not (voxel_size > 0.0):
        raise RuntimeError("[Open3D Error] [VoxelDownSample] voxel_size <= 0.")
    L = _lib.load(require_device=True)
    px, kx, dx, n = _prep32(xyz, 3, device)
    pc, kc, dc, nc = _prep32(cov6, 6, device)
    pk, kk, dk, nk = _prep32(color, 3, device)
    for d, m, name in ((dc, nc, "cov6"), (dk, nk, "color")):
        if d is not None and (d != dx or m != n):
            raise RuntimeError(f"voxel_down_sample: {name} must match xyz in length and placement")
    if dx and torch is not None:
        torch.cuda.current_stream(device).synchronize()
    h = C.c_void_p()
    nv = C.c_int64(0)
    stream = torch.cuda.current_stream(device).cuda_stream if (torch is not None and torch.cuda.is_available()) else 0
    try:
        _lib.check(L.gsr_voxel_down_sample(int(device), C.c_void_p(stream), px, pc, pk, n, float(voxel_size), 1 if dx else 0,
                                           C.byref(h), C.byref(nv)), "gsr_voxel_down_sample")
        V = int(nv.value)
        if as_torch:
            dev = torch.device("cuda", device)
            ox = torch.empty((V, 3), dtype=torch.float64, device=dev)
            oc = torch.empty((V, 6), dtype=torch.float64, device=dev) if cov6 is not None else None
            ok = torch.empty((V, 3), dtype=torch.float64, device=dev) if color is not None else None
            ptr = lambda t: None if t is None else t.data_ptr()
            _lib.check(L.gsr_voxel_fetch(h, ptr(ox), ptr(oc), ptr(ok), 1), "gsr_voxel_fetch")
        else:
            ox = np.empty((V, 3), np.float64)
            oc = np.empty((V, 6), np.float64) if cov6 is not None else None
            ok = np.empty((V, 3), np.float64) if color is not None else None
            ptr = lambda a: None if a is None else a.ctypes.data
            _lib.check(L.gsr_voxel_fetch(h, ptr(ox), ptr(oc), ptr(ok), 0), "gsr_voxel_fetch")
        return ox, oc, ok
    finally:
        if h:
            L.gsr_voxel_free(h)
