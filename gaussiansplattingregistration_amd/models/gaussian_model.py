"""Splat tensor container with the accessor names of the reference's ``GaussianModel``
(``src/models/gaussian_model.py:21``) -- the part of it the hot path touches.

Kept (same names / shapes): ``get_xyz (N,3)`` ``:56-57``, ``get_colors (N,3)`` ``:66-67``,
``get_spherical_harmonics (N, 3*((deg+1)^2-1))`` ``:70-71``, ``get_raw_opacity (N,1)`` ``:78-79``,
``get_covariance(1) (N,6)`` ``:89-91``, ``get_full_covariance()`` ``:81-87``, ``from_mixture(model, sh_degree)``
``:141-153``, ``move_to_device`` ``:223-234``, ``clone_gaussian``, ``from_ply`` / ``save_ply`` (``:98-139,169-185``; own
reader/writer in ``utils/ply_io.py``, ``plyfile`` is not needed).  ``from_arrays`` builds level 0 from raw arrays.

``from_mixture`` runs the reference's scaling/rotation rebuild (``:151-153,242-265``: batched ``eigh``, axis matching,
quaternions; the reference's own comment calls it unused) only when asked (``decompose=True`` / ``"reference"`` /
``"exact"``): registration only consumes xyz / covariance.  The rebuild is one device kernel (``csrc/model.hip``).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .gaussian_mixture_level import GaussianMixtureModel


def _t(a, device, shape=None):
    if isinstance(a, torch.Tensor):
        t = a.detach().to(device=device, dtype=torch.float32)
    else:
        t = torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)
    return t.reshape(shape) if shape is not None else t


def _matrices_to_quaternions(R):
    """(N,3,3) -> (N,4) (w, x, y, z), the trace formula the reference uses (``general_utils.py:94-100``; no branch for
    w -> 0, as there)."""
    w = torch.sqrt(1.0 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) * 0.5
    d = 4.0 * w
    return torch.stack((w, (R[:, 2, 1] - R[:, 1, 2]) / d, (R[:, 0, 2] - R[:, 2, 0]) / d, (R[:, 1, 0] - R[:, 0, 1]) / d), dim=-1)


class GaussianModel:
    def __init__(self, device_name="cpu"):
        self.sh_degree = -1
        self.device_name = device_name
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self._covariance = torch.empty(0)

    # -- accessors (reference names) -------------------------------------------------------------
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)                              # gaussian_model.py:48-50

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)          # gaussian_model.py:52-54

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_colors(self):
        return self._features_dc.flatten(start_dim=1)

    @property
    def get_spherical_harmonics(self):
        return self._features_rest.flatten(start_dim=1)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_raw_opacity(self):
        return self._opacity

    @property
    def get_opacity_with_activation(self):
        return torch.sigmoid(self._opacity)

    def get_full_covariance(self, scaling_modifier=1.0):
        c = self._covariance
        full = torch.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], dim=1)
        if scaling_modifier == 1:
            return full
        return full * float(scaling_modifier) ** 2

    def get_covariance(self, scaling_modifier=1):
        if scaling_modifier == 1:
            return self._covariance
        # the reference applies the diagonal scaling twice (gaussian_model.py:93-96): S (S C S^T) S^T
        return self._covariance * float(scaling_modifier) ** 4

    def __len__(self):
        return int(self._xyz.shape[0])

    # -- construction ------------------------------------------------------------------------------
    def from_arrays(self, xyz, colors, opacities, covariance, features, sh_degree):
        """Level 0 from the five arrays the HEM boundary takes (opacity RAW)."""
        n = int(xyz.shape[0])
        self.sh_degree = sh_degree
        k = (sh_degree + 1) ** 2 - 1
        self._xyz = _t(xyz, self.device_name, (n, 3))
        self._features_dc = _t(colors, self.device_name, (n, 1, 3))
        self._features_rest = _t(features, self.device_name, (n, k, 3))
        self._opacity = _t(opacities, self.device_name, (n, 1))
        self._covariance = _t(covariance, self.device_name, (n, 6))
        return self

    def from_ply(self, path_or_arrays, device=None, timing=None):
        """``GaussianModel.from_ply`` (reference ``gaussian_model.py:98-139``) from a file path (own reader,
        ``utils/ply_io.py``) or from the dict ``ply_io.load_gaussian_arrays`` returns.  ``device`` (a CUDA index; default: this
        model's ``device_name`` when that is a CUDA device): the file goes through pinned chunks straight into device SoA
        (``ply_io.load_gaussian_device``) -- the reference, too, loads onto ``cuda:0`` (``file_loader.py:53-66``)."""
        from ..utils import ply_io
        is_path = isinstance(path_or_arrays, (str, bytes)) or hasattr(path_or_arrays, "__fspath__")
        if device is None and is_path and str(self.device_name).startswith("cuda"):
            device = torch.device(self.device_name).index or 0
        if is_path and device is not None:
            self.device_name = f"cuda:{int(device)}"
            try:
                d = ply_io.load_gaussian_device(path_or_arrays, int(device), timing=timing)
            except ValueError:
                # the device loader takes binary little-endian float32 files whose vertex element comes first (what 3DGS writes); anything else the
                # format allows -- ASCII, big-endian, other property types, other element orders -- goes through the host reader, like the
                # reference's plyfile would read it, and is uploaded afterwards (from_arrays below)
                d = ply_io.load_gaussian_arrays(path_or_arrays)
        else:
            d = ply_io.load_gaussian_arrays(path_or_arrays) if is_path else path_or_arrays
        self.from_arrays(d["xyz"], d["color"], d["opacity"], d["cov6"], d["sh"], d["sh_degree"])
        self._scaling = _t(d["scale"], self.device_name)
        self._rotation = _t(d["rot"], self.device_name)
        return self

    def save_ply(self, path):
        """``GaussianModel.save_ply`` (``gaussian_model.py:169-185``)."""
        from ..utils import ply_io
        if self._scaling.numel() == 0:
            raise RuntimeError("save_ply needs scaling/rotation: build the model with from_ply or from_mixture(..., decompose=True)")
        c = lambda t: t.detach().cpu().numpy()
        ply_io.save_gaussian_ply(path, c(self._xyz), c(self.get_colors), c(self.get_spherical_harmonics), c(self._opacity),
                                 c(self._scaling), c(self._rotation))

    def from_mixture(self, gaussian_mixture: GaussianMixtureModel, sh_degree: int, decompose=False):
        """``GaussianModel.from_mixture`` (``gaussian_model.py:141-153``).  ``decompose``: ``False`` skips the scaling /
        rotation rebuild (registration consumes xyz / covariance only); ``True`` or ``"reference"`` runs it with the
        reference's arithmetic, ``"exact"`` with a decomposition that really reproduces the covariance (what
        ``save_ply`` of a down-sampled model needs) -- both on the GPU (``csrc/model.hip``)."""
        self.sh_degree = sh_degree
        k = (sh_degree + 1) ** 2 - 1
        self._xyz = _t(gaussian_mixture.xyz, self.device_name)
        n = int(self._xyz.shape[0])
        self._features_dc = _t(gaussian_mixture.colors, self.device_name).view(n, 1, 3)
        self._features_rest = _t(gaussian_mixture.features, self.device_name).view(n, k, 3)
        self._opacity = _t(gaussian_mixture.opacities, self.device_name)
        self._covariance = _t(gaussian_mixture.covariance, self.device_name).view(n, 6)
        if decompose:
            mode = "exact" if decompose == "exact" else "reference"
            if mode == "reference":
                self._scaling, evec = self.decompose_covariance_matrix()
                self._rotation = self._last_quaternions
            else:
                self._scaling, self._rotation, _ = self._decompose(_lib.GSR_DECOMP_EXACT)
        return self

    def _decompose(self, mode):
        """-> (scaling (N,3), quaternions (N,4), matrices (N,3,3)) of ``gsr_decompose_cov`` on the model's covariances."""
        import ctypes as C
        L = _lib.load(require_device=True)
        cov = self._covariance.detach().to(torch.float32).contiguous()
        n = int(cov.shape[0])
        if cov.is_cuda:
            sc = torch.empty((n, 3), dtype=torch.float32, device=cov.device)
            q = torch.empty((n, 4), dtype=torch.float32, device=cov.device)
            mat = torch.empty((n, 3, 3), dtype=torch.float32, device=cov.device)
            torch.cuda.current_stream(cov.device.index).synchronize()
            _lib.check(L.gsr_decompose_cov(cov.data_ptr(), n, mode, sc.data_ptr(), q.data_ptr(), mat.data_ptr(), 1, cov.device.index,
                                           C.c_void_p(torch.cuda.current_stream(cov.device.index).cuda_stream)), "gsr_decompose_cov")
            return sc, q, mat
        a = np.ascontiguousarray(cov.numpy())
        sc, q, mat = np.empty((n, 3), np.float32), np.empty((n, 4), np.float32), np.empty((n, 3, 3), np.float32)
        _lib.check(L.gsr_decompose_cov(a.ctypes.data, n, mode, sc.ctypes.data, q.ctypes.data, mat.ctypes.data, 0, 0, None), "gsr_decompose_cov")
        return torch.from_numpy(sc), torch.from_numpy(q), torch.from_numpy(mat)

    def decompose_covariance_matrix(self):
        """Scaling / rotation of every component from its covariance with the reference's arithmetic
        (``gaussian_model.py:242-265``), in ONE device kernel (``gsr_decompose_cov``, ``GSR_DECOMP_REFERENCE``) instead of
        a batched ``torch.linalg.eigh`` plus scatters: eigenpair k goes to the slot of the coordinate axis its eigenvector
        is most aligned with; two claims of one slot overwrite in eigenvalue order, an unclaimed slot stays zero.
        Returns (values (N,3), vectors (N,3,3)); like the reference's, the "scaling" is the eigenvalue itself, not its
        square root or logarithm.  The quaternions of the same call are kept in ``_last_quaternions``."""
        sc, q, mat = self._decompose(_lib.GSR_DECOMP_REFERENCE)
        self._last_quaternions = q
        return sc, mat

    # -- rigid motion and merge (reference ``gaussian_model.py:198-222,267-290``) ---------------------------------
    def transform_gaussian_model(self, transformation_matrix):
        """Apply a rigid 4x4 to positions, covariances and rotation quaternions, in place.  (SH coefficients are left
        as they are, as in the reference.)"""
        T = torch.as_tensor(transformation_matrix, dtype=torch.float32, device=self._xyz.device)
        R, t = T[:3, :3], T[:3, 3]
        self._xyz = self._xyz @ R.T + t
        n = len(self)
        C = self.get_full_covariance()
        # R C R^T as two (3n x 3) GEMMs (a batched matmul with > 2^24 batches faults on this ROCm build)
        D = (C.reshape(n * 3, 3) @ R.T).reshape(n, 3, 3)
        C = (D.transpose(1, 2).reshape(n * 3, 3) @ R.T).reshape(n, 3, 3).transpose(1, 2)
        self._covariance = torch.stack([C[:, 0, 0], C[:, 0, 1], C[:, 0, 2], C[:, 1, 1], C[:, 1, 2], C[:, 2, 2]], dim=1)
        if self._rotation.numel():
            qr = _matrices_to_quaternions(R[None])[0]                       # (w, x, y, z) of the motion
            w0, x0, y0, z0 = self._rotation.unbind(-1)
            w1, x1, y1, z1 = qr
            q = torch.stack((w1 * w0 - x1 * x0 - y1 * y0 - z1 * z0,          # Hamilton product, reference operand order (:198-208)
                             w1 * x0 + x1 * w0 + y1 * z0 - z1 * y0,
                             w1 * y0 - x1 * z0 + y1 * w0 + z1 * x0,
                             w1 * z0 + x1 * y0 - y1 * x0 + z1 * w0), dim=-1)
            self._rotation = q / torch.linalg.vector_norm(q, dim=-1, keepdim=True)
        return self

    @staticmethod
    def get_merged_gaussian_point_clouds(gaussian1, gaussian2, transformation_matrix):
        """``gaussian1`` moved by the registration result, concatenated with ``gaussian2`` (the merged-cloud save)."""
        g1 = gaussian1
        if transformation_matrix is not None and not np.array_equal(np.asarray(transformation_matrix), np.eye(4)):
            g1 = gaussian1.clone_gaussian()
            g1.transform_gaussian_model(np.asarray(transformation_matrix, dtype=np.float32))
        assert gaussian1.sh_degree == gaussian2.sh_degree
        m = GaussianModel(gaussian2.device_name)
        m.sh_degree = gaussian1.sh_degree
        for name in ("_xyz", "_rotation", "_scaling", "_features_dc", "_features_rest", "_opacity", "_covariance"):
            setattr(m, name, torch.cat((getattr(g1, name).to(gaussian2.device_name), getattr(gaussian2, name))))
        return m

    def clone_gaussian(self):
        m = GaussianModel(self.device_name)
        m.sh_degree = self.sh_degree
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_covariance"):
            setattr(m, name, getattr(self, name).clone().detach())
        return m

    def move_to_device(self, device_name):
        if self.device_name == device_name:
            return
        self.device_name = device_name
        for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "_covariance"):
            setattr(self, name, getattr(self, name).to(device_name))
