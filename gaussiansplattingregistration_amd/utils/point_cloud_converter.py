"""``convert_gs_to_open3d_pc`` of the reference (``src/utils/point_cloud_converter.py:31-49``) without
Open3D and without the host round trip the reference's README calls too slow (``README.md:115``):
positions, SH-DC colours (``sh2rgb``), packed covariances and covariance-derived normals stay on the
device the model lives on."""
from __future__ import annotations

from ..models.point_cloud import PointCloud
from .graphics_utils import sh2rgb


def convert_input_pc_to_open3d_pc(vertices):
    """Sparse input cloud (``x y z red green blue`` vertices, e.g. COLMAP's ``points3D.ply``) -> ``PointCloud`` with
    colours / 255 and KNN-30 normals (reference ``point_cloud_converter.py:9-28``).  ``vertices`` is the structured array
    ``ply_io.read_ply_vertices`` returns (the reference passes the ``plyfile`` object)."""
    import numpy as np
    xyz = np.stack([vertices["x"], vertices["y"], vertices["z"]], 1).astype(np.float32)
    colors = np.stack([vertices["red"], vertices["green"], vertices["blue"]], 1).astype(np.float64) / 255
    pc = PointCloud(xyz32=xyz, colors=colors)
    pc.estimate_normals()
    return pc


def convert_gs_to_open3d_pc(gaussian):
    pc = PointCloud(xyz32=gaussian.get_xyz.detach(), colors=sh2rgb(gaussian.get_colors.detach().double()),
                    cov6=gaussian.get_covariance(1).detach())
    pc.estimate_normals()
    return pc
