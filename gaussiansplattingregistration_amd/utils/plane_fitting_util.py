"""RANSAC plane fitting of the reference (``src/utils/plane_fitting_util.py:6-99``) with the same functions, arguments and
return values; the per-candidate work over all points -- five full-size torch kernels per RANSAC iteration on the CPU
there -- is ONE pass of a HIP kernel over the points for all ``iterations`` candidates of a plane (``gsr_plane_score``,
``csrc/model.hip``).

What stays on the host, as in the reference: drawing the three sample points of every candidate from torch's global
generator (``torch.randint`` in the reference's order, so a seeded run draws the same candidates), the minimum-distance
rejection between them, the candidate's normal and offset (float32 torch ops, bit for bit the reference's).  The samples
of all iterations are drawn first -- sampling never looks at the scores -- then scored together.

After a plane's inliers are removed the reference filters ``points_tensor`` but keeps indexing the UNFILTERED normals with the
filtered indices (``:15,23-27,57``), so from the second plane on it tests the normals of the wrong points.
``fit_planes(..., reference_compat=True)`` reproduces exactly that (point k of the filtered set is tested against normal k of the
ORIGINAL cloud; golden vectors from the reference's own function: ``tests/golden/planes2.npz``); the default filters the
normals with the points -- the intended behaviour.  The first plane is identical either way.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib


def sample_random_points(points: torch.Tensor, min_distance: float) -> torch.Tensor:
    """``sample_random_points`` (``:72-93``): three indices, each new one at least ``min_distance`` from the chosen ones."""
    num_points = points.shape[0]
    selected_indices = [torch.randint(0, num_points, (1,)).item()]
    while len(selected_indices) < 3:
        candidate_index = torch.randint(0, num_points, (1,)).item()
        distances = torch.norm(points[selected_indices] - points[candidate_index], dim=1)
        if torch.all(distances >= min_distance):
            selected_indices.append(candidate_index)
    return torch.tensor(selected_indices)


def project_point_onto_plane(points, plane):
    """``project_point_onto_plane`` (``:96-105``) in torch (host); used for inspection, the scoring kernel inlines it."""
    a, b, c, d = plane
    normal = torch.tensor([a, b, c], dtype=torch.float32)
    normal = normal / normal.norm()
    distances = (torch.mm(points, normal.unsqueeze(1)).squeeze() + d) / normal.norm()
    projected_points = points - distances.unsqueeze(1) * normal
    return projected_points, distances


def _candidate(points_tensor, sampled_indices):
    """Plane through three points as ``_fit_single_plane`` builds it (``:44-50``) + the constants the kernel needs."""
    sampled_points = points_tensor[sampled_indices]
    p1, p2, p3 = sampled_points[0], sampled_points[1], sampled_points[2]
    normal = torch.cross(p2 - p1, p3 - p1, dim=0)
    normal /= normal.norm()
    d = -torch.dot(normal, p1)
    plane = np.hstack((normal.numpy(), d.numpy()))
    n2 = torch.tensor([plane[0], plane[1], plane[2]], dtype=torch.float32)     # project_point_onto_plane re-normalises
    n2 = n2 / n2.norm()
    return plane, np.float32([n2[0], n2[1], n2[2], plane[3], normal[0], normal[1], normal[2], n2.norm()])


def _fit_single_plane(points_tensor, normal_tensors, iterations, distance_threshold, normal_threshold, min_sample_distance, device=0,
                      points_dev=None, normals_dev=None):
    """``_fit_single_plane`` (``:38-69``): best plane (4 floats) and the indices of its inliers, or (None, None)."""
    if iterations <= 0 or points_tensor.shape[0] == 0:
        return None, None
    planes, cands = [], np.empty((iterations, 8), np.float32)
    for it in range(iterations):
        plane, cands[it] = _candidate(points_tensor, sample_random_points(points_tensor, min_sample_distance))
        planes.append(plane)
    L = _lib.load(require_device=True)
    n = int(points_tensor.shape[0])
    counts = np.zeros(iterations, np.uint32)
    best = C.c_int32(-1)
    if points_dev is not None:
        mask = torch.empty(n, dtype=torch.uint8, device=points_dev.device)
        torch.cuda.current_stream(points_dev.device.index).synchronize()
        _lib.check(L.gsr_plane_score(points_dev.data_ptr(), normals_dev.data_ptr(), n, cands.ctypes.data, iterations, float(distance_threshold),
                                     float(normal_threshold), counts.ctypes.data, mask.data_ptr(), C.byref(best), 1, points_dev.device.index,
                                     C.c_void_p(torch.cuda.current_stream(points_dev.device.index).cuda_stream)), "gsr_plane_score")
        mask_t = mask
    else:
        pts = np.ascontiguousarray(points_tensor.numpy(), dtype=np.float32)
        nrm = np.ascontiguousarray(normal_tensors.numpy(), dtype=np.float32)
        mask = np.empty(n, np.uint8)
        _lib.check(L.gsr_plane_score(pts.ctypes.data, nrm.ctypes.data, n, cands.ctypes.data, iterations, float(distance_threshold),
                                     float(normal_threshold), counts.ctypes.data, mask.ctypes.data, C.byref(best), 0, int(device), None), "gsr_plane_score")
        mask_t = torch.from_numpy(mask)
    if best.value < 0:
        return None, None
    return planes[best.value], mask_t.nonzero(as_tuple=True)[0].cpu().numpy()


def fit_planes(point_cloud, plane_count, iterations, threshold, normal_threshold, min_sample_distance, device=None, reference_compat=False):
    """``fit_planes`` (``:6-35``): up to ``plane_count`` planes, each the RANSAC winner over the points the earlier planes
    left; returns (list of plane coefficient arrays [a, b, c, d], list of ORIGINAL index tensors of their inliers).
    ``point_cloud`` needs ``points`` and ``normals`` (the ``PointCloud`` record or an Open3D cloud).
    ``reference_compat``: index the unfiltered normals with the filtered indices from the second plane on, as the reference does."""
    points_tensor = torch.from_numpy(np.array(point_cloud.points, dtype=np.float32))
    nrm = point_cloud.normals
    nrm = nrm.detach().cpu().numpy() if isinstance(nrm, torch.Tensor) else np.asarray(nrm)
    normal_tensor = torch.from_numpy(np.array(nrm, dtype=np.float32))
    dev = device if device is not None else getattr(point_cloud, "device_index", 0)
    original_indices = torch.arange(points_tensor.shape[0])
    plane_coefficients, inlier_indices_list = [], []
    pd = nd = None
    normal_all = normal_tensor                          # reference_compat: the normals are never filtered
    if torch.cuda.is_available():                       # keep the (shrinking) point set resident between planes
        pd, nd = points_tensor.to(f"cuda:{dev}").contiguous(), normal_tensor.to(f"cuda:{dev}").contiguous()
    nd_all = nd
    for _ in range(plane_count):
        best_plane, best_inliers = _fit_single_plane(points_tensor, normal_tensor, iterations, threshold, normal_threshold,
                                                     min_sample_distance, dev, pd, nd)
        if best_plane is not None:
            plane_coefficients.append(best_plane)
            inlier_indices_list.append(original_indices[best_inliers])
            mask = torch.ones(points_tensor.shape[0], dtype=torch.bool)
            mask[best_inliers] = False
            points_tensor = points_tensor[mask]
            # the reference forgets to filter the normals (see the module docstring): point k of the filtered set then meets
            # normal k of the original cloud
            normal_tensor = normal_all[: points_tensor.shape[0]] if reference_compat else normal_tensor[mask]
            original_indices = original_indices[mask]
            if pd is not None:
                md = mask.to(pd.device)
                pd = pd[md].contiguous()
                nd = nd_all[: pd.shape[0]].contiguous() if reference_compat else nd[md].contiguous()
        else:
            break
        if points_tensor.shape[0] == 0:
            break
    return plane_coefficients, inlier_indices_list
