#!/usr/bin/env python3
"""Golden vectors for the RANSAC plane search: fit_planes of the REFERENCE
(/root/reference/src/utils/plane_fitting_util.py:6-69), executed in the build container on the CPU with a seeded torch
generator.

    python tests/golden/make_golden_planes.py        -> tests/golden/planes.npz

The module imports open3d at its top for a type annotation and for get_o3d_plane (a mesh helper, not called here); open3d
is not installed, so an empty placeholder module object lets the `import` succeed.  fit_planes / _fit_single_plane /
sample_random_points / project_point_onto_plane run unmodified.  planes.npz: plane_count = 1.  planes2.npz: plane_count = 3 --
from the second plane on the reference indexes the unfiltered normals with filtered indices; the repo reproduces that with
fit_planes(..., reference_compat=True) (gaussiansplattingregistration_amd/utils/plane_fitting_util.py).  The fixtures hold
inputs and outputs only.
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
if "open3d" not in sys.modules:
    ph = types.ModuleType("open3d")
    ph.geometry = types.SimpleNamespace(PointCloud=object)
    sys.modules["open3d"] = ph
from src.utils.plane_fitting_util import fit_planes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from plane_scene import scene  # noqa: E402


def main():
    out = {}
    for case, (seed, iters, thr, nthr, msd) in enumerate(((5, 120, 0.02, 0.9, 0.3), (9, 60, 0.05, 0.8, 0.5))):
        pts, nrm = scene(seed)
        pc = types.SimpleNamespace(points=pts, normals=nrm)
        torch.manual_seed(1000 + case)
        planes, inliers = fit_planes(pc, 1, iters, thr, nthr, msd)
        out.update({f"points_{case}": pts.astype(np.float32), f"normals_{case}": nrm.astype(np.float32),
                    f"params_{case}": np.array([iters, thr, nthr, msd, 1000 + case], np.float64),
                    f"plane_{case}": np.asarray(planes[0], np.float32), f"inliers_{case}": inliers[0].numpy().astype(np.int64)})
        print("case", case, "plane", planes[0], "inliers", len(inliers[0]))
    np.savez_compressed(os.path.join(HERE, "planes.npz"), torch_version=np.array(torch.__version__), **out)
    # several planes: the reference's behaviour from the second plane on (unfiltered normals)
    out = {}
    pts, nrm = scene(31)
    pc = types.SimpleNamespace(points=pts, normals=nrm)
    torch.manual_seed(2024)
    planes, inliers = fit_planes(pc, 3, 100, 0.02, 0.9, 0.3)
    out.update(points=pts.astype(np.float32), normals=nrm.astype(np.float32), params=np.array([3, 100, 0.02, 0.9, 0.3, 2024], np.float64),
               n_planes=np.array(len(planes)))
    for k, (pl, il) in enumerate(zip(planes, inliers)):
        out[f"plane_{k}"] = np.asarray(pl, np.float32)
        out[f"inliers_{k}"] = il.numpy().astype(np.int64)
        print("planes2", k, pl, len(il))
    np.savez_compressed(os.path.join(HERE, "planes2.npz"), torch_version=np.array(torch.__version__), threads=np.array(torch.get_num_threads()), **out)


if __name__ == "__main__":
    main()
