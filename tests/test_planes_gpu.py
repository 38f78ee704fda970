"""N4: RANSAC plane fitting and plane-inlier HEM merging.

fit_planes against golden vectors produced by the REFERENCE's own fit_planes (tests/golden/planes.npz,
make_golden_planes.py) under the same torch seed: the candidates are drawn on the host with the reference's torch calls, so
the winning plane must be the same plane, and so must its inlier set.  The reference's distances come out of torch.mm on the
CPU (MKL sgemv); the scoring kernel rounds its three-term dot products the way MKL's main loop does --
fl(z n2) + fma(y, n1, fl(x n0)), measured in the build container to reproduce torch.mm bit for bit except on the rows in the tail
of each OpenMP thread's share (about 1e-4 of the rows with 8 threads; which rows depends on the machine's thread count, so no
device code can follow it).  A tail row flips only if its distance also sits within an ulp of the threshold: the fixtures'
inlier sets come out identical."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


class _PC:
    def __init__(self, points, normals):
        self.points, self.normals = points, normals


@pytest.mark.parametrize("case", [0, 1])
def test_fit_planes_against_reference_vectors(case):
    from gaussiansplattingregistration_amd.utils.plane_fitting_util import fit_planes
    g = np.load(os.path.join(GOLDEN, "planes.npz"))
    iters, thr, nthr, msd, seed = g[f"params_{case}"]
    torch.manual_seed(int(seed))
    planes, inliers = fit_planes(_PC(g[f"points_{case}"], g[f"normals_{case}"]), 1, int(iters), float(thr), float(nthr), float(msd))
    assert len(planes) == 1 and np.array_equal(planes[0], g[f"plane_{case}"])          # the same candidate wins, bit for bit
    assert np.array_equal(inliers[0].numpy(), g[f"inliers_{case}"]), (len(inliers[0]), len(g[f"inliers_{case}"]))
    assert inliers[0].dtype == torch.int64 and bool((inliers[0][1:] > inliers[0][:-1]).all())


def test_fit_planes_reference_compat_several_planes():
    """plane_count = 3 as the REFERENCE computes it (tests/golden/planes2.npz, its own fit_planes): from the second plane on it
    tests the filtered points against the UNFILTERED normals (plane_fitting_util.py:15,23-27,57), which leaves planes two and
    three with ~100 inliers each.  reference_compat=True reproduces it: same planes bit for bit, same inlier index sets.  The
    default (normals filtered with the points) finds the planes the scene really has."""
    from gaussiansplattingregistration_amd.utils.plane_fitting_util import fit_planes
    g = np.load(os.path.join(GOLDEN, "planes2.npz"))
    count, iters, thr, nthr, msd, seed = g["params"]
    torch.manual_seed(int(seed))
    planes, inliers = fit_planes(_PC(g["points"], g["normals"]), int(count), int(iters), float(thr), float(nthr), float(msd), reference_compat=True)
    assert len(planes) == int(g["n_planes"]) == 3
    for k in range(3):
        assert np.array_equal(planes[k], g[f"plane_{k}"]), k
        assert np.array_equal(inliers[k].numpy(), g[f"inliers_{k}"]), (k, len(inliers[k]), len(g[f"inliers_{k}"]))
    torch.manual_seed(int(seed))
    planes2, inliers2 = fit_planes(_PC(g["points"], g["normals"]), int(count), int(iters), float(thr), float(nthr), float(msd))
    assert np.array_equal(planes2[0], planes[0]) and np.array_equal(inliers2[0].numpy(), inliers[0].numpy())     # the first plane: identical
    assert len(inliers2[1]) > 10 * len(inliers[1])                                                                # then the intended behaviour


def test_fit_planes_several_planes_and_exhaustion():
    """Three planes one after the other: every plane found once, inliers disjoint, original indices; and the device scoring
    equals a float32 numpy restatement of the same test on every candidate."""
    import sys
    from gaussiansplattingregistration_amd.utils import plane_fitting_util as pf
    sys.path.insert(0, GOLDEN)
    from plane_scene import scene
    pts, nrm = scene(21)
    torch.manual_seed(77)
    planes, inl = pf.fit_planes(_PC(pts, nrm), 3, 150, 0.02, 0.9, 0.3)
    assert len(planes) == 3
    sizes = sorted(len(i) for i in inl)
    assert sizes[2] > 5500 and sizes[1] > 3000 and sizes[0] > 2000
    allidx = torch.cat(inl)
    assert len(set(allidx.tolist())) == len(allidx)                                   # disjoint
    p32, n32 = pts.astype(np.float32), nrm.astype(np.float32)
    for plane, idx in zip(planes, inl):
        d = np.abs(p32[idx.numpy()] @ plane[:3] + plane[3])
        assert d.max() < 0.02 * 1.001 and (np.abs(n32[idx.numpy()] @ plane[:3]) > 0.9 * 0.999).all()
    # scoring kernel against numpy on every candidate of one search
    torch.manual_seed(5)
    pt = torch.from_numpy(p32)
    cands = np.stack([pf._candidate(pt, pf.sample_random_points(pt, 0.3))[1] for _ in range(40)])
    import ctypes as C
    from gaussiansplattingregistration_amd import _lib
    L = _lib.load(require_device=True)
    counts = np.zeros(40, np.uint32); best = C.c_int32(-1); mask = np.empty(len(p32), np.uint8)
    assert L.gsr_plane_score(p32.ctypes.data, n32.ctypes.data, len(p32), cands.ctypes.data, 40, 0.02, 0.9, counts.ctypes.data, mask.ctypes.data,
                             C.byref(best), 0, 0, None) == 0
    want = []
    for c in cands:
        dist = (p32 @ c[:3] + c[3]) / c[7]
        want.append(int(((np.abs(dist) < np.float32(0.02)) & (np.abs(n32 @ c[4:7]) > np.float32(0.9))).sum()))
    assert np.abs(counts.astype(np.int64) - np.array(want)).max() <= 3                # float32 summation order of the dot products
    assert best.value == int(np.argmax(counts)) and mask.sum() == counts[best.value]
    # nothing to find: no plane, empty lists
    torch.manual_seed(1)
    assert pf.fit_planes(_PC(pts[:50], nrm[:50]), 2, 20, 1e-9, 0.999999, 0.0) == ([], [])


def test_plane_inlier_merging_worker():
    """PlaneInlierMergingWorker: per level = unselected components + the HEM levels of every plane's inliers, in the
    reference's order and on its shared rand() stream (first cloud's planes, then the second cloud's)."""
    from gaussiansplattingregistration_amd import hem, mixture_bind, synth
    from gaussiansplattingregistration_amd.models.gaussian_model import GaussianModel
    from gaussiansplattingregistration_amd.params import GaussianMixtureParams
    from gaussiansplattingregistration_amd.workers.plane_merging import PlaneInlierMergingWorker
    clouds = [synth.make_cloud(6000, seed=s, h=0.8, sh_degree=1) for s in (1, 2)]
    models = [GaussianModel("cuda:0").from_arrays(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], 1) for c in clouds]
    rng = np.random.default_rng(0)
    perm = [rng.permutation(6000) for _ in range(2)]
    planes = [[np.sort(p[:2500]), np.sort(p[2500:4000])] for p in perm]               # two "planes" per cloud, 2000 unselected
    params = GaussianMixtureParams(cluster_level=2)
    mixture_bind.reset_rng()
    res = PlaneInlierMergingWorker(models[0], models[1], planes[0], planes[1], params).run()
    assert len(res.list_gaussian_first) == 2 and len(res.list_open3d_second) == 2
    # the same by hand: HEM of every subset with the stream position carried from subset to subset
    pos = 0
    for ci, out in ((0, res.list_gaussian_first), (1, res.list_gaussian_second)):
        c = clouds[ci]
        unsel = np.setdiff1d(np.arange(6000), np.concatenate(planes[ci]))
        want = [[c["xyz"][unsel]] for _ in range(2)]
        for idx in planes[ci]:
            sub = {k: c[k][idx] for k in ("xyz", "color", "opacity", "cov6", "sh")}
            with hem.HemMixture(rng_mode="glibc", rng_seed=1, rng_skip=pos) as m:
                m.set_level0(sub["xyz"], sub["color"], sub["opacity"], sub["cov6"], sub["sh"])
                for k in range(2):
                    m.run_level()
                    want[k].append(m.get_level()["xyz"])
                pos = m.stats()["rng_draws"]
        for k in range(2):
            w = np.concatenate(want[k])
            assert out[k].get_xyz.shape[0] == w.shape[0]
            assert np.array_equal(out[k].get_xyz.cpu().numpy(), w)
            assert len(res.list_open3d_first[k]) == res.list_gaussian_first[k].get_xyz.shape[0]
            assert out[k].sh_degree == 1 and out[k].get_spherical_harmonics.shape[1] == 9
