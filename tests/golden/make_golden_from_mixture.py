#!/usr/bin/env python3
"""Golden vectors for the level-export glue: GaussianModel.decompose_covariance_matrix and
matrices_to_quaternions of the REFERENCE, executed in the build container on the CPU.

    python tests/golden/make_golden_from_mixture.py        -> tests/golden/from_mixture.npz

The two functions are imported from /root/reference (src/models/gaussian_model.py:242-265, src/utils/general_utils.py:94-100)
and run unmodified.  `plyfile` -- imported at the top of gaussian_model.py for from_ply / save_ply only -- is not
installed in this image: an empty placeholder module object is registered so that the `import` statement succeeds; no
plyfile function is reached by the two functions executed here.  Nothing of the reference is copied: the fixture holds
inputs (cov6) and outputs (arrays) only.

Cases: random anisotropic splats, near-axis-aligned ones, isotropic / repeated eigenvalues, and covariances whose two
largest-|component| eigenvectors claim the same axis (the scatter_ overwrite case; torch's CPU scatter_ lets the later
index win).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
if "plyfile" not in sys.modules:
    ph = types.ModuleType("plyfile")
    ph.PlyElement = ph.PlyData = None
    sys.modules["plyfile"] = ph
from src.models.gaussian_model import GaussianModel  # noqa: E402
from src.utils.general_utils import matrices_to_quaternions  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rot(q):
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.empty((q.shape[0], 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def main():
    rng = np.random.default_rng(2024)
    n = 1500
    s = np.exp(rng.normal(-2.5, 0.7, (n, 3)))
    q = rng.normal(size=(n, 4))
    q[500:700] = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.02, (200, 4))          # near axis-aligned
    s[700:760] = s[700:760, :1]                                                    # isotropic
    s[760:820, 1] = s[760:820, 0]                                                  # a repeated eigenvalue
    # rotations about (1,1,1) by ~60 degrees mix the axes evenly: frequent double claims of one axis
    ang = np.deg2rad(rng.uniform(40, 80, 300))
    q[900:1200] = np.concatenate([np.cos(ang / 2)[:, None], np.sin(ang / 2)[:, None] * np.ones((300, 3)) / np.sqrt(3)], 1)
    R = rot(q)
    L = R * s[:, None, :]
    C = (L @ L.transpose(0, 2, 1)).astype(np.float32)
    cov6 = np.ascontiguousarray(C[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]])
    g = GaussianModel("cpu")
    g._covariance = torch.tensor(cov6)
    vals, vecs = g.decompose_covariance_matrix()
    quat = matrices_to_quaternions(vecs)
    ev, evec = torch.linalg.eigh(g.get_full_covariance())
    corr = torch.argmax(torch.abs(evec.transpose(1, 2)), dim=2)
    np.savez_compressed(os.path.join(HERE, "from_mixture.npz"), cov6=cov6, sorted_eigenvalues=vals.numpy(), sorted_eigenvectors=vecs.numpy(),
                        quaternions=quat.numpy(), eigenvalues=ev.numpy(), eigenvectors=evec.numpy(), correspondence=corr.numpy(),
                        torch_version=np.array(torch.__version__), numpy_version=np.array(np.__version__))
    dbl = int((np.sort(corr.numpy(), 1)[:, 1:] == np.sort(corr.numpy(), 1)[:, :-1]).any(1).sum())
    print("wrote from_mixture.npz:", n, "covariances,", dbl, "with a doubly claimed axis")


if __name__ == "__main__":
    main()
