"""Light point-cloud record standing in for ``o3d.geometry.PointCloud`` on the ICP path.

The reference hands Open3D clouds to ``do_icp_registration``; Open3D is an un-vendored wheel, so the
backend uses this record instead: ``points`` (N,3) float64 view, ``colors``, ``covariances`` (N,3,3),
``normals`` (N,3) float64.  The float32 coordinates (``xyz32`` -- what the splats really store,
``point_cloud_converter.py:33`` only widens them) are the primary storage and may be a PyTorch-ROCm
tensor, in which case the ICP kernels read them in place.
"""
from __future__ import annotations

import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(a):
    return torch is not None and isinstance(a, torch.Tensor)


class PointCloud:
    def __init__(self, xyz32=None, colors=None, cov6=None, normals=None):
        self.xyz32 = xyz32 if xyz32 is not None else np.zeros((0, 3), np.float32)
        self.colors = colors
        self.cov6 = cov6
        self.normals = normals

    @property
    def device_index(self):
        if _is_tensor(self.xyz32) and self.xyz32.is_cuda:
            return self.xyz32.device.index
        return 0

    def __len__(self):
        return int(self.xyz32.shape[0])

    def __repr__(self):
        return f"PointCloud with {len(self)} points."

    @property
    def points(self):
        if _is_tensor(self.xyz32):
            return self.xyz32.detach().double().cpu().numpy()
        return np.asarray(self.xyz32, dtype=np.float64)

    @property
    def covariances(self):
        if self.cov6 is None:
            return None
        c = self.cov6.detach().double().cpu().numpy() if _is_tensor(self.cov6) else np.asarray(self.cov6, np.float64)
        return np.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], axis=1)

    def has_normals(self):
        return self.normals is not None and self.normals.shape[0] == len(self) and len(self) > 0

    def has_covariances(self):
        return self.cov6 is not None

    def estimate_normals(self, knn=30):
        """Open3D ``estimate_normals()``, on the GPU.  A cloud whose covariances are set (every splat cloud,
        ``point_cloud_converter.py:40-43``) gets the smallest-eigenvalue eigenvector of each covariance; a cloud
        without covariances (a sparse input cloud, ``point_cloud_converter.py:9-28``) gets Open3D's default: the
        covariance of each point's ``knn`` = 30 nearest neighbours."""
        from .. import icp
        if self.cov6 is None:
            self.normals = icp.normals_knn(self.xyz32, knn=knn, device=self.device_index)
        else:
            self.normals = icp.normals_from_cov(self.cov6, device=self.device_index)
        return self

    def voxel_down_sample(self, voxel_size):
        """``o3d.geometry.PointCloud.voxel_down_sample``: one point per occupied voxel = the mean of the voxel's points,
        covariances and colours (float64 means, kept here in the record's float32 storage; voxels in ascending index
        order).  Runs on the GPU (``csrc/voxel.hip``)."""
        from .. import voxel
        xyz, cov6, col = voxel.voxel_down_sample(self.xyz32, voxel_size, cov6=self.cov6, color=self.colors, device=self.device_index,
                                                 as_torch=_is_tensor(self.xyz32) and self.xyz32.is_cuda)
        f32 = (lambda a: None if a is None else (a.float() if _is_tensor(a) else a.astype(np.float32)))
        return PointCloud(xyz32=f32(xyz), colors=f32(col), cov6=f32(cov6))

    def transform(self, T):
        T = np.asarray(T, dtype=np.float64)
        p = self.points @ T[:3, :3].T + T[:3, 3]
        self.xyz32 = p.astype(np.float32)
        if self.normals is not None:
            n = self.normals.detach().cpu().numpy() if _is_tensor(self.normals) else self.normals
            self.normals = n @ T[:3, :3].T
        return self
