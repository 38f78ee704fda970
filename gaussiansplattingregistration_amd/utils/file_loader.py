"""File loading for the registration workers (reference ``src/utils/file_loader.py:14-80``): the point-cloud type of a
``.ply`` by its vertex properties, sparse input clouds, 3DGS clouds -- with the own reader of ``ply_io`` instead of
``plyfile`` / Open3D."""
from __future__ import annotations

import os.path
from enum import IntEnum, auto

from . import ply_io
from .point_cloud_converter import convert_gs_to_open3d_pc, convert_input_pc_to_open3d_pc


class PointCloudType(IntEnum):
    GAUSSIAN = auto()
    INPUT = auto()
    UNKNOWN = auto()


def check_point_cloud_type(vertices):
    """``vertices``: the structured vertex array, or just the tuple of its property names (the header is enough)."""
    props = vertices if isinstance(vertices, (tuple, list)) else (vertices.dtype.names or ())
    if "red" in props:
        return PointCloudType.INPUT
    if "f_dc_0" in props:
        return PointCloudType.GAUSSIAN
    return PointCloudType.UNKNOWN


def load_sparse_pc(pc_path):
    """``load_sparse_pc`` (``file_loader.py:20-30``): None unless the file exists and is an INPUT-type cloud."""
    if not pc_path or not os.path.isfile(pc_path):
        return None
    vertices = ply_io.read_ply_vertices(pc_path)
    if check_point_cloud_type(vertices) is not PointCloudType.INPUT:
        return None
    return convert_input_pc_to_open3d_pc(vertices)


def load_gaussian_pc(pc_path, device_name="cuda:0"):
    """``load_gaussian_pc`` (``file_loader.py:53-66``): (point cloud, GaussianModel) or (None, None)."""
    from ..models.gaussian_model import GaussianModel
    if not pc_path or not os.path.isfile(pc_path):
        return None, None
    if check_point_cloud_type(ply_io.read_ply_property_names(pc_path)) is not PointCloudType.GAUSSIAN:       # the header decides
        return None, None
    g = GaussianModel(device_name).from_ply(pc_path)          # a CUDA device: pinned chunks straight into device SoA (ply_io.load_gaussian_device)
    return convert_gs_to_open3d_pc(g), g
