"""ICP parity proper (HIP path through the C ABI) against the oracle and the committed fixture.
Tolerance (BASELINE.json north_star): final transform within 1e-5 Frobenius."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL_T = 1e-5


def _cov3(c6):
    c = c6.astype(np.float64)
    return np.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], 1)


@pytest.mark.parametrize("name,kind,loss,k", [("p2p", 0, 0, 0.0), ("p2plane", 1, 0, 0.0), ("p2plane_tukey", 1, 1, 0.05),
                                              ("p2plane_huber", 1, 4, 0.01)])
def test_gpu_vs_icp_fixture(name, kind, loss, k):
    from gaussiansplattingregistration_amd import icp
    g = load_golden("icp_pair")
    r = icp.registration_icp_arrays(g["src_xyz"], g["tgt_xyz"], g["tgt_normals"], np.eye(4), kind=kind, loss=loss, k=k,
                                    max_corr=float(g["max_corr"]), max_iter=int(g["max_iter"]))
    assert r["iterations"] == int(g[f"{name}_iters"])
    assert np.linalg.norm(r["transformation"] - g[f"{name}_T"]) < TOL_T
    assert abs(r["fitness"] - float(g[f"{name}_fitness"])) < 1e-9 and abs(r["inlier_rmse"] - float(g[f"{name}_rmse"])) < 1e-8
    assert np.linalg.norm(r["transformation"] - g["T_gt"]) < 5e-3


@pytest.mark.parametrize("kind,loss,k", [(0, 0, 0.0), (1, 0, 0.0), (1, 2, 0.05), (1, 3, 0.05)])
def test_gpu_vs_oracle_100k(oracle, kind, loss, k):
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(100000, seed=5, sh_degree=0, angle_deg=2.0)
    nrm = icp.normals_from_cov(tgt["cov6"])
    assert np.abs(np.abs((nrm * oracle.normals_from_cov(_cov3(tgt["cov6"]))).sum(1)) - 1).max() < 1e-9
    init = np.eye(4)
    init[:3, 3] = [0.01, -0.02, 0.005]
    w = oracle.icp(src["xyz"], tgt["xyz"], nrm, init, kind=kind, loss=loss, k=k, max_corr=0.2, max_iter=15)
    r = icp.registration_icp_arrays(src["xyz"], tgt["xyz"], nrm, init, kind=kind, loss=loss, k=k, max_corr=0.2, max_iter=15)
    assert r["iterations"] == w["iterations"]
    assert np.linalg.norm(r["transformation"] - w["transformation"]) < TOL_T
    assert abs(r["fitness"] - w["fitness"]) < 1e-9 and abs(r["inlier_rmse"] - w["inlier_rmse"]) < 1e-8


def test_correspondences_exact(oracle):
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(50000, seed=7, sh_degree=0)
    T = np.eye(4)
    T[:3, 3] = 0.03
    with icp.IcpContext() as c:
        c.set_target(tgt["xyz"], None, 0.15)
        c.set_source(src["xyz"])
        idx, d2 = c.correspondences(T)
    widx, wd2 = oracle.icp_correspond(src["xyz"], tgt["xyz"], T, 0.15)
    assert np.array_equal(idx, widx)                           # index work: bit-exact
    assert np.allclose(d2, wd2, rtol=1e-12, atol=0)
    assert (idx < 0).any() and (idx >= 0).any()                # both outcomes exercised


@pytest.mark.parametrize("case", ["offset", "ties", "lattice", "outside", "sparse_far", "clustered"])
def test_correspondences_exact_adversarial(oracle, case):
    """The grid search does its geometry in float32 with slack and only the near-best candidates in float64;
    its answers must still be the exact float64 nearest neighbours with ties to the lowest index: large coordinate
    offsets, duplicated target points, queries on cell boundaries, queries far outside the target box, a max_corr
    much larger than the cell, strongly clustered targets."""
    from gaussiansplattingregistration_amd import icp
    rng = np.random.default_rng(17)
    n = 20000
    tgt = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    src = (tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    max_corr = 0.1
    T = np.eye(4)
    if case == "offset":                       # float32 coordinates around 2000: 1.2e-4 spacing
        tgt = tgt + np.float32(2000.0); src = src + np.float32(2000.0)
    elif case == "ties":                       # every target point four times; sources AT target points
        tgt = np.concatenate([tgt[:5000]] * 4); src = tgt[rng.permutation(len(tgt))[:n]].copy()
    elif case == "lattice":                    # targets and sources on a lattice: equal distances and cell faces everywhere
        g = np.stack(np.meshgrid(*[np.arange(28)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.0625)
        tgt = g; src = (g[rng.permutation(len(g))[:n]] + np.float32(0.03125)).astype(np.float32)
    elif case == "outside":                    # half of the sources far outside the target box
        src[: n // 2] = (src[: n // 2] * 30.0).astype(np.float32); max_corr = 50.0
    elif case == "sparse_far":                 # few targets, huge max_corr (many rings)
        tgt = tgt[:300]; max_corr = 3.0
    elif case == "clustered":
        tgt[: n // 2] = (tgt[: n // 2] * 0.01).astype(np.float32)
        T[:3, 3] = [0.01, -0.02, 0.005]
    with icp.IcpContext() as c:
        c.set_target(tgt, None, max_corr)
        c.set_source(src)
        idx, d2 = c.correspondences(T)
    widx, wd2 = oracle.icp_correspond(src, tgt, T, max_corr)
    assert np.array_equal(idx, widx), (case, int((idx != widx).sum()))
    assert np.allclose(d2, wd2, rtol=1e-12, atol=0)
    assert (idx >= 0).any()


@pytest.mark.parametrize("case", ["plain", "offset", "ties", "lattice", "outside", "clustered", "moved"])
def test_tile_search_of_a_fine_level_is_exact(monkeypatch, oracle, case):
    """k_icp_nn_tile: on a fine level (the correspondence distance is at most one grid cell: rings == 1) a workgroup stages the target
    points of the box around its 256 queries in LDS and every lane scans its 27 cells from there.  GSR_ICP_TILE=1 routes
    gsr_icp_correspondences through it: the indices must be the oracle's KD-tree neighbours bit for bit -- duplicated target points
    (ties to the lowest index), coordinates around 2000, queries on cell faces, queries far outside the target's box (clamped cells),
    a clustered target (boxes that do not fit the tile fall back to the per-thread search), a source that moved by several cells since
    it was sorted (boxes too large: fallback again)."""
    from gaussiansplattingregistration_amd import icp
    monkeypatch.setenv("GSR_ICP_TILE", "1")
    rng = np.random.default_rng(23)
    n = 60000
    tgt = rng.uniform(-1.6, 1.6, (n, 3)).astype(np.float32)
    src = (tgt[rng.permutation(n)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
    max_corr = 0.1                              # ~ one cell at two points per cell
    T = np.eye(4)
    T[:3, 3] = [0.01, -0.02, 0.005]
    if case == "offset":
        tgt = tgt + np.float32(2000.0); src = src + np.float32(2000.0)
    elif case == "ties":
        tgt = np.concatenate([tgt[:15000]] * 4); src = tgt[rng.permutation(len(tgt))[:n]].copy(); T = np.eye(4)
    elif case == "lattice":
        g = np.stack(np.meshgrid(*[np.arange(40)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.0625)
        tgt = g; src = (g[rng.permutation(len(g))[:n]] + np.float32(0.03125)).astype(np.float32); max_corr = 0.06; T = np.eye(4)
    elif case == "outside":
        src[: n // 2] = (src[: n // 2] * 3.0).astype(np.float32)
    elif case == "clustered":
        tgt[: n // 2] = (tgt[: n // 2] * 0.02).astype(np.float32)
    elif case == "moved":                        # the evaluation's transform is far from the one the source was sorted under
        T[:3, 3] = [0.9, -0.7, 0.4]
    with icp.IcpContext() as c:
        c.set_target(tgt, None, max_corr)
        c.set_source(src)
        idx, d2 = c.correspondences(T)
        idx0, d20 = c.correspondences(np.eye(4))
    for TT, (gi, gd) in ((T, (idx, d2)), (np.eye(4), (idx0, d20))):
        widx, wd2 = oracle.icp_correspond(src, tgt, TT, max_corr)
        assert np.array_equal(gi, widx), (case, int((gi != widx).sum()))
        assert np.allclose(gd, wd2, rtol=1e-12, atol=0)
    assert (idx >= 0).any() or (idx0 >= 0).any()


def test_tile_search_changes_no_registration(monkeypatch):
    """A registration whose finest schedule entry searches over LDS tiles (the default from 400 k source points on when max_corr is at
    most a cell) against the same with GSR_ICP_TILE=0 (every level searched per thread): transform, fitness, RMSE and iteration count
    equal -- the neighbours are the same, the accumulate kernel does not know who found them."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, _ = synth.make_pair(450000, seed=14, sh_degree=0)
    nrm = icp.normals_from_cov(tgt["cov6"])
    res = {}
    for tile in ("0", "-1"):
        monkeypatch.setenv("GSR_ICP_TILE", tile)
        out = []
        for kind, nn in ((0, None), (1, nrm)):
            r = icp.registration_icp_arrays(src["xyz"], tgt["xyz"], nn, np.eye(4), kind=kind, max_corr=0.07, max_iter=12)
            out.append((r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"]))
        res[tile] = out
    for a, b in zip(res["0"], res["-1"]):
        assert np.abs(a[0] - b[0]).max() < 1e-12 and a[3] == b[3] and abs(a[1] - b[1]) < 1e-15 and abs(a[2] - b[2]) < 1e-12
    assert res["0"][0][1] > 0.5


def test_accumulators_match_numpy():
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, _ = synth.make_pair(20000, seed=8, sh_degree=0)
    with icp.IcpContext() as c:
        c.set_target(tgt["xyz"], None, 0.2)
        c.set_source(src["xyz"])
        acc = c.accumulate(np.eye(4), kind=0)
        idx, d2 = c.correspondences(np.eye(4))
    m = idx >= 0
    assert acc[0] == m.sum() and abs(acc[1] - d2[m].sum()) < 1e-9 * d2[m].sum()


def test_do_icp_registration_and_multiscale_driver(oracle):
    """Reference-shaped call path: GaussianModel -> HEM worker -> converter -> multiscale ICP (intended 10-arg form),
    against the oracle chained the same way."""
    import torch
    from gaussiansplattingregistration_amd import mixture_bind, synth
    from gaussiansplattingregistration_amd.controllers.downsampler_controller import DownsamplerController
    from gaussiansplattingregistration_amd.controllers.registration_controller import RegistrationController
    from gaussiansplattingregistration_amd.models.data_repository import DataRepository, UIStateRepository
    from gaussiansplattingregistration_amd.models.gaussian_model import GaussianModel
    from gaussiansplattingregistration_amd.params import GaussianMixtureParams
    from gaussiansplattingregistration_amd.utils.local_registration_util import KernelLossFunctionType, LocalRegistrationType
    from gaussiansplattingregistration_amd.utils.point_cloud_converter import convert_gs_to_open3d_pc

    src, tgt, T_gt = synth.make_pair(20000, seed=11, sh_degree=1, angle_deg=3.0)
    repo, ui = DataRepository(), UIStateRepository()
    for c, gl, ol in ((src, repo.pc_gaussian_list_first, repo.pc_open3d_list_first), (tgt, repo.pc_gaussian_list_second, repo.pc_open3d_list_second)):
        gm = GaussianModel("cuda:0").from_arrays(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], 1)
        gl.append(gm)
        ol.append(convert_gs_to_open3d_pc(gm))
    mixture_bind.reset_rng()
    DownsamplerController(repo).create_mixture(GaussianMixtureParams(cluster_level=2))
    assert len(repo.pc_open3d_list_first) == 3 and len(repo.pc_gaussian_list_second) == 3      # [orig, L1, L2]
    rc = RegistrationController(repo, ui)
    voxel, iters = [0.6, 0.4, 0.2], [20, 15, 10]
    res = rc.execute_multiscale_registration(False, "", "", LocalRegistrationType.ICP_Point_To_Plane, 1e-6, 1e-6, voxel, iters,
                                             KernelLossFunctionType.Loss_None, 0.0, True)
    assert res is not None, rc.errors
    T = res.result.transformation
    assert np.array_equal(ui.transformation_matrix, T)
    # oracle chain on the very same level clouds (coarse -> fine)
    Tw = np.eye(4)
    for k in range(3):
        s, t = repo.pc_open3d_list_first[-(k + 1)], repo.pc_open3d_list_second[-(k + 1)]
        nrm = t.normals.cpu().numpy() if isinstance(t.normals, torch.Tensor) else t.normals
        w = oracle.icp(s.points, t.points, nrm, Tw, kind=1, max_corr=voxel[k], max_iter=iters[k])
        Tw = w["transformation"]
    assert np.linalg.norm(T - Tw) < TOL_T
    assert np.linalg.norm(T - T_gt) < 0.05


def _cov33(c6):
    c6 = np.asarray(c6, np.float64)
    return np.stack([c6[:, [0, 1, 2]], c6[:, [1, 3, 4]], c6[:, [2, 4, 5]]], 1)


@pytest.mark.parametrize("loss,k,n", [(0, 0.0, 5000), (1, 0.5, 5000), (4, 0.3, 5000), (0, 0.0, 150000)])
def test_generalized_icp_vs_oracle(oracle, loss, k, n):
    """registration_generalized_icp with the splats' own covariances (reference local_registration_util.py:96-98):
    GPU against the oracle (itself cross-checked against NumPy/SciPy), L2 / Tukey / Huber, small and large."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(n, seed=13, sh_degree=0)
    r = icp.registration_icp_arrays(src["xyz"], tgt["xyz"], None, np.eye(4), kind=2, loss=loss, k=k, max_corr=0.3, max_iter=25,
                                    src_cov=src["cov6"], tgt_cov=tgt["cov6"])
    w = oracle.gicp(src["xyz"], _cov33(src["cov6"]), tgt["xyz"], _cov33(tgt["cov6"]), np.eye(4), loss=loss, k=k, max_corr=0.3,
                    max_iter=25)
    assert r["iterations"] == w["iterations"]
    assert np.linalg.norm(r["transformation"] - w["transformation"]) < TOL_T
    assert abs(r["fitness"] - w["fitness"]) < 1e-9 and abs(r["inlier_rmse"] - w["inlier_rmse"]) < 1e-8
    if n <= 5000:                                                     # (the big scene's offset needs more than 25 iterations)
        assert np.linalg.norm(r["transformation"] - T_gt) < 0.05


def test_generalized_icp_through_do_icp_registration(oracle):
    """The reference-shaped call: LocalRegistrationType.ICP_General on converted clouds (device tensors, non-identity init)."""
    from gaussiansplattingregistration_amd import synth
    from gaussiansplattingregistration_amd.models.gaussian_model import GaussianModel
    from gaussiansplattingregistration_amd.params.registration_parameters import LocalRegistrationParams
    from gaussiansplattingregistration_amd.utils.local_registration_util import (KernelLossFunctionType, LocalRegistrationType,
                                                                                 do_icp_registration)
    from gaussiansplattingregistration_amd.utils.point_cloud_converter import convert_gs_to_open3d_pc
    from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
    src, tgt, T_gt = synth.make_pair(8000, seed=14, sh_degree=0)
    pcs = []
    for c in (src, tgt):
        gm = GaussianModel("cuda:0").from_arrays(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], 0)
        pcs.append(convert_gs_to_open3d_pc(gm))
    init = synth.rigid_transform(1.0, (0, 0, 1), (0.01, 0.0, -0.01))
    p = LocalRegistrationParams(registration_type=LocalRegistrationType.ICP_General, max_correspondence=0.3, relative_fitness=1e-6,
                                relative_rmse=1e-6, max_iteration=20, rejection_type=KernelLossFunctionType.Cauchy_Loss, k_value=0.4)
    res = do_icp_registration(pcs[0], pcs[1], init, p)
    w = oracle.gicp(src["xyz"], _cov33(src["cov6"]), tgt["xyz"], _cov33(tgt["cov6"]), init, loss=2, k=0.4, max_corr=0.3, max_iter=20)
    assert np.linalg.norm(res.transformation - w["transformation"]) < TOL_T
    assert abs(res.fitness - w["fitness"]) < 1e-9
    # a cloud WITHOUT covariances (the reference's sparse input clouds reach this estimator through
    # qt_multiscale_registrator.py:82-85): Open3D's InitializePointCloudForGeneralizedICP gives it discs of thickness 1e-3
    # perpendicular to its normals -- the normals it carries, or 20-nearest-neighbour normals when it has none
    from gaussiansplattingregistration_amd import icp
    nrm = oracle.normals_knn(src["xyz"].astype(np.float64), 30)
    got = icp.cov_from_normals(nrm, 1e-3)
    want = oracle.cov_from_normals(nrm, 1e-3)
    assert np.abs(got - want[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]).max() < 1e-15
    flip = np.array([[-1.0, 0.0, 0.0], [-0.995, 0.0998, 0.0], [1.0, 0.0, 0.0]])          # c < -0.99: the identity, as Open3D has it
    assert np.abs(icp.cov_from_normals(flip) - oracle.cov_from_normals(flip)[:, [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]).max() < 1e-15
    for variant in ("normals", "bare"):
        a = PointCloud(xyz32=src["xyz"], normals=nrm if variant == "normals" else None)       # no covariances
        res = do_icp_registration(a, pcs[1], init, p)
        n_used = nrm if variant == "normals" else oracle.normals_knn(src["xyz"].astype(np.float64), 20)
        w = oracle.gicp(src["xyz"], oracle.cov_from_normals(n_used, 1e-3), tgt["xyz"], _cov33(tgt["cov6"]), init, loss=2, k=0.4, max_corr=0.3, max_iter=20)
        assert np.linalg.norm(res.transformation - w["transformation"]) < TOL_T, variant
        assert abs(res.fitness - w["fitness"]) < 1e-9 and res.iterations == w["iterations"]


def _colored_pair(n, seed, oracle):
    from test_icp_oracle import _colored_pair as cp
    return cp(n, seed, oracle)


@pytest.mark.parametrize("n,radius", [(4000, 0.6), (4000, 0.09), (120000, 0.3)])
def test_color_gradient_vs_oracle(oracle, n, radius):
    """InitializePointCloudForColoredICP on the GPU: the 30 nearest neighbours within the radius, ordered by (distance,
    index), and the tangent-plane least-squares gradient -- same neighbours, same summation order as the oracle."""
    from gaussiansplattingregistration_amd import icp
    s, scol, t, nrm, tcol, _ = _colored_pair(n, 15, oracle)
    with icp.IcpContext() as c:
        c.set_target(t.astype(np.float32), nrm, radius / 2.0)          # gradients use Hybrid(2 * max_corr, 30)
        c.set_target_color(tcol)
        got = c.color_gradient()
    want = oracle.color_gradient(t.astype(np.float32).astype(np.float64), nrm, tcol, radius)
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() < 1e-9 * scale
    if radius < 0.1:
        assert (np.abs(want).sum(1) == 0).any()                      # points with fewer than 4 neighbours keep a zero gradient


@pytest.mark.parametrize("loss,k", [(0, 0.0), (1, 0.4), (2, 0.3)])
def test_colored_icp_vs_oracle(oracle, loss, k):
    from gaussiansplattingregistration_amd import icp
    s, scol, t, nrm, tcol, T_gt = _colored_pair(6000, 16, oracle)
    s32, t32 = s.astype(np.float32), t.astype(np.float32)
    r = icp.registration_icp_arrays(s32, t32, nrm, np.eye(4), kind=3, loss=loss, k=k, max_corr=0.3, max_iter=25, src_color=scol,
                                    tgt_color=tcol)
    w = oracle.colored_icp(s32.astype(np.float64), scol, t32.astype(np.float64), nrm, tcol, np.eye(4), loss=loss, k=k, max_corr=0.3,
                           max_iter=25)
    assert r["iterations"] == w["iterations"]
    assert np.linalg.norm(r["transformation"] - w["transformation"]) < TOL_T
    assert abs(r["fitness"] - w["fitness"]) < 1e-9 and abs(r["inlier_rmse"] - w["inlier_rmse"]) < 1e-8
    assert np.linalg.norm(r["transformation"] - T_gt) < 0.05


def test_colored_icp_through_do_icp_registration(oracle):
    from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
    from gaussiansplattingregistration_amd.params.registration_parameters import LocalRegistrationParams
    from gaussiansplattingregistration_amd.utils.local_registration_util import (KernelLossFunctionType, LocalRegistrationType,
                                                                                 do_icp_registration)
    s, scol, t, nrm, tcol, T_gt = _colored_pair(5000, 17, oracle)
    src = PointCloud(xyz32=s.astype(np.float32), colors=scol)
    tgt = PointCloud(xyz32=t.astype(np.float32), colors=tcol, normals=nrm)
    p = LocalRegistrationParams(registration_type=LocalRegistrationType.ICP_Color, max_correspondence=0.25, relative_fitness=1e-6,
                                relative_rmse=1e-6, max_iteration=20, rejection_type=KernelLossFunctionType.Loss_None, k_value=0.0)
    res = do_icp_registration(src, tgt, np.eye(4), p)
    w = oracle.colored_icp(s.astype(np.float32).astype(np.float64), scol, t.astype(np.float32).astype(np.float64), nrm, tcol, np.eye(4),
                           max_corr=0.25, max_iter=20)
    assert np.linalg.norm(res.transformation - w["transformation"]) < TOL_T
    tgt.colors = None
    with pytest.raises(RuntimeError, match="color"):
        do_icp_registration(src, tgt, np.eye(4), p)


@pytest.mark.parametrize("on_device", [False, True])
@pytest.mark.parametrize("kind", [0, 1])
def test_registration_in_one_call_equals_the_step_by_step_entry_points(kind, on_device):
    """gsr_icp_register_clouds -- Open3D's registration_icp(source, target, max_corr, init, estimation, criteria) signature
    (local_registration_util.py:88-90): index build, source sort and the loop without a stream synchronisation or a return to Python between
    them -- against gsr_icp_set_target + _set_source + _register on the same context: the same bits, the same iteration count, the timing
    filled in; a context that held a sharded call's callback before is clean afterwards; precondition errors are the step-by-step ones."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(60000, seed=11)
    nrm = icp.normals_from_cov(tgt["cov6"])
    sx, tx, tn = src["xyz"], tgt["xyz"], np.asarray(nrm)
    if on_device:
        sx, tx, tn = torch.from_numpy(sx).cuda(), torch.from_numpy(tx).cuda(), torch.from_numpy(tn).cuda()
    with icp.IcpContext() as c:
        c.set_target(tx, tn if kind == 1 else None, 0.3)
        c.set_source(sx)
        want = c.register(np.eye(4), kind, 0, 0.0, 1e-6, 1e-6, 40)
        t_want = c.timing()
        for _ in range(2):
            got = c.register_clouds(sx, tx, tn if kind == 1 else None, 0.3, np.eye(4), kind, 0, 0.0, 1e-6, 1e-6, 40)
            t_got = c.timing()
            assert got["iterations"] == want["iterations"] and np.array_equal(got["transformation"], want["transformation"])
            assert got["fitness"] == want["fitness"] and got["inlier_rmse"] == want["inlier_rmse"]
            assert t_got["ms_build"] > 0 and t_got["iter_kernels"] == t_want["iter_kernels"]
        assert np.linalg.norm(got["transformation"] - T_gt) < 5e-3
        with pytest.raises(RuntimeError, match="max_correspondence_distance"):
            c.register_clouds(sx, tx, tn if kind == 1 else None, 0.0, np.eye(4), kind)
        if kind == 1:
            with pytest.raises(RuntimeError, match="requires target normals"):
                c.register_clouds(sx, tx, None, 0.3, np.eye(4), 1)
        got = c.register_clouds(sx, tx, tn if kind == 1 else None, 0.3, np.eye(4), kind, 0, 0.0, 1e-6, 1e-6, 40)       # the context still works
        assert np.array_equal(got["transformation"], want["transformation"])


@pytest.mark.parametrize("knob", ["GSR_ICP_DEVICE_LOOP=0", "GSR_ICP_NN_KERNEL=2", "GSR_ICP_XCD=0", "GSR_ICP_RB_POLL=0", "GSR_ICP_ROBUST_BOX=0"])
def test_registration_in_one_call_under_the_knobs(monkeypatch, knob):
    """gsr_icp_register_clouds under the library's own knobs (the host-driven loop keeps its stream waits: defer_sync follows device_loop): the result of
    the default run to 1e-12, the same iteration count."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(50000, seed=19)
    nrm = np.asarray(icp.normals_from_cov(tgt["cov6"]))
    with icp.IcpContext() as c:
        want = c.register_clouds(src["xyz"], tgt["xyz"], nrm, 0.3, np.eye(4), 1, 0, 0.0, 1e-6, 1e-6, 40)
    k, v = knob.split("=")
    monkeypatch.setenv(k, v)
    with icp.IcpContext() as c:
        got = c.register_clouds(src["xyz"], tgt["xyz"], nrm, 0.3, np.eye(4), 1, 0, 0.0, 1e-6, 1e-6, 40)
        assert c.timing()["ms_build"] > 0
    assert got["iterations"] == want["iterations"] and np.linalg.norm(got["transformation"] - want["transformation"]) < 1e-12, knob


def test_coarse_to_fine_schedule_in_one_call_equals_the_entry_by_entry_loop():
    """gsr_icp_register_multiscale (MultiScaleRegistratorMixture._register_main_point_clouds, qt_multiscale_registrator.py:197-236) against the loop of
    gsr_icp_register_clouds it replaces: every entry's start, transform, fitness, RMSE and iteration count equal bit for bit; an empty schedule returns
    the initial transform."""
    from gaussiansplattingregistration_amd import icp, synth
    rng = np.random.default_rng(3)
    src, tgt, T_gt = synth.make_pair(80000, seed=13)
    nrm = np.asarray(icp.normals_from_cov(tgt["cov6"]))
    sub = [np.sort(rng.choice(80000, m, replace=False)) for m in (5000, 20000)] + [np.arange(80000)]
    entries = [(np.ascontiguousarray(src["xyz"][i]), np.ascontiguousarray(tgt["xyz"][i]), np.ascontiguousarray(nrm[i]), mc, it)
               for i, mc, it in zip(sub, (0.6, 0.4, 0.2), (30, 20, 10))]
    for on_device in (False, True):
        ent = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c_).cuda(), mc, it) for a, b, c_, mc, it in entries] if on_device else entries
        with icp.IcpContext() as c:
            want, T = [], np.eye(4)
            for sx, tx, tn, mc, it in ent:
                r = c.register_clouds(sx, tx, tn, mc, T, 1, 0, 0.0, 1e-6, 1e-6, it)
                want.append((T.copy(), r))
                T = r["transformation"]
            got = c.register_multiscale(ent, np.eye(4), 1, 0, 0.0, 1e-6, 1e-6)
            assert len(got) == 3
            for (T0, w), g in zip(want, got):
                assert np.array_equal(g["init"], T0) and np.array_equal(g["transformation"], w["transformation"])
                assert g["iterations"] == w["iterations"] and g["fitness"] == w["fitness"] and g["inlier_rmse"] == w["inlier_rmse"]
                assert g["ms_build"] > 0 and g["evaluations"] >= g["iterations"]
            assert np.linalg.norm(got[-1]["transformation"] - T_gt) < 5e-3
            assert c.register_multiscale([], np.eye(4)) == []


def test_icp_error_behaviour():
    from gaussiansplattingregistration_amd import icp
    with icp.IcpContext() as c:
        with pytest.raises(RuntimeError, match="max_correspondence_distance"):
            c.set_target(np.zeros((4, 3), np.float32), None, 0.0)
        c.set_target(np.random.rand(10, 3).astype(np.float32), None, 0.5)
        c.set_source(np.random.rand(10, 3).astype(np.float32))
        with pytest.raises(RuntimeError, match="normals"):
            c.register(kind=1)


def test_new_entry_points_error_behaviour():
    """The round-6 entry points through the raw C ABI: NULL / negative arguments and unsupported estimators are refused with a message, nothing crashes,
    and the context serves a good call afterwards."""
    import ctypes as C
    from gaussiansplattingregistration_amd import _lib, hem, icp, synth
    L = _lib.load()
    T0 = np.eye(4)
    out = np.empty((4, 4))
    fit, rm, it = C.c_double(0), C.c_double(0), C.c_int32(0)
    xyz = np.random.default_rng(0).random((200, 3)).astype(np.float32)
    with icp.IcpContext() as c:
        call = lambda *a: L.gsr_icp_register_clouds(c._h, *a, out.ctypes.data, C.byref(fit), C.byref(rm), C.byref(it))
        assert call(xyz.ctypes.data, 200, xyz.ctypes.data, None, 200, 0, 0.3, T0.ctypes.data, 2, 0, 0.0, 1e-6, 1e-6, 5) < 0           # generalized ICP: not this entry
        assert b"point-to-point and point-to-plane" in L.gsr_last_error()
        assert call(xyz.ctypes.data, 200, xyz.ctypes.data, None, 200, 0, 0.3, T0.ctypes.data, 1, 0, 0.0, 1e-6, 1e-6, 5) < 0           # point-to-plane without normals
        assert call(xyz.ctypes.data, 0, xyz.ctypes.data, None, 200, 0, 0.3, T0.ctypes.data, 0, 0, 0.0, 1e-6, 1e-6, 5) < 0             # empty source
        assert call(xyz.ctypes.data, 200, None, None, 0, 0, 0.3, T0.ctypes.data, 0, 0, 0.0, 1e-6, 1e-6, 5) < 0                        # empty target
        assert call(xyz.ctypes.data, 200, xyz.ctypes.data, None, 200, 0, -1.0, T0.ctypes.data, 0, 0, 0.0, 1e-6, 1e-6, 5) < 0          # max_corr <= 0
        assert L.gsr_icp_register_multiscale(c._h, 2, None, 0, T0.ctypes.data, 0, 0, 0.0, 1e-6, 1e-6, None, out.ctypes.data) < 0
        assert L.gsr_icp_register_multiscale(c._h, -1, None, 0, T0.ctypes.data, 0, 0, 0.0, 1e-6, 1e-6, None, out.ctypes.data) < 0
        r = c.register_clouds(xyz, xyz, None, 0.3, T0, 0, 0, 0.0, 1e-6, 1e-6, 5)                                                       # ... and a good call
        assert r["fitness"] == 1.0 and np.allclose(r["transformation"], np.eye(4), atol=1e-12)
    cloud = synth.make_cloud(2000, seed=1, sh_degree=1, h=0.4)
    with hem.HemMixture() as m:
        reports = (_lib.HemLevelReport * 2)()
        rp = C.cast(reports, C.c_void_p)
        assert L.gsr_hem_run_levels(m._h, 1, None, None, None, None, None, None, None, 100, rp) < 0                                   # no level set
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        assert L.gsr_hem_run_levels(m._h, 1, None, None, None, None, None, None, None, 100, rp) < 0                                   # NULL arenas
        assert L.gsr_hem_run_levels(m._h, -1, None, None, None, None, None, None, None, 100, rp) < 0
        a = m.new_arena(4096)
        assert L.gsr_hem_run_levels(m._h, 1, a["xyz"].data_ptr(), a["color"].data_ptr(), a["cov6"].data_ptr(), a["opacity"].data_ptr(), None, None, None, 4096, rp) < 0
        assert b"no SH arena" in L.gsr_last_error()
        assert L.gsr_hem_run_levels(m._h, 2, a["xyz"].data_ptr(), a["color"].data_ptr(), a["cov6"].data_ptr(), a["opacity"].data_ptr(), a["sh"].data_ptr(), None, None, 4096, None) < 0
        levels, st = m.run_levels(2, arena=a)                                                                                          # ... and a good call
        assert st[0]["n_in"] == 2000 and levels[1]["xyz"].shape[0] == st[1]["n_out"]


def test_icp_knobs_change_nothing(monkeypatch):
    """The environment knobs of the ICP half select implementations, never results: the fused search + accumulate kernel
    against the split one (GSR_ICP_NN_KERNEL), the host-driven loop against the device-resident one (GSR_ICP_DEVICE_LOOP),
    other grid resolutions (GSR_ICP_CELL_TARGET, GSR_ICP_MAX_CELLS), the ring loop from ring 0 against the 27-cell block of
    batched loads the coarse levels start with (GSR_ICP_BLOCK_SEARCH; max_corr = 0.3 spans several cells here, and the
    split-kernel case runs it in k_icp_nn too), another number of workgroups -- another association of the points with the partial
    sums (GSR_ICP_BLOCKS) --, workgroups without the XCD-contiguous mapping of the source ranges (GSR_ICP_XCD=0), k_icp_step as its own launch against the accumulate kernel's last workgroup doing
    its work (GSR_ICP_FUSED_STEP: one launch per iteration instead of two)."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, _ = synth.make_pair(50000, seed=13, sh_degree=0)
    C = tgt["cov6"]
    nrm = icp.normals_from_cov(C)

    def run():
        out = []
        for kind, n in ((0, None), (1, nrm)):
            r = icp.registration_icp_arrays(src["xyz"], tgt["xyz"], n, np.eye(4), kind=kind, max_corr=0.3, max_iter=20)
            out.append((r["transformation"], r["fitness"], r["inlier_rmse"], r["iterations"]))
        with icp.IcpContext() as c:
            c.set_target(tgt["xyz"], None, 0.3)
            c.set_source(src["xyz"])
            out.append(c.correspondences(np.eye(4)))
        return out

    ref = run()
    for env in ({"GSR_ICP_NN_KERNEL": "0"}, {"GSR_ICP_NN_KERNEL": "2"}, {"GSR_ICP_DEVICE_LOOP": "0"}, {"GSR_ICP_CELL_TARGET": "0.5"},
                {"GSR_ICP_CELL_TARGET": "16"}, {"GSR_ICP_MAX_CELLS": "4096"}, {"GSR_ICP_BLOCK_SEARCH": "0"},
                {"GSR_ICP_BLOCKS": "100"}, {"GSR_ICP_RB_POLL": "0"}, {"GSR_ICP_XCD": "0"},
                {"GSR_ICP_BLOCK_SEARCH": "0", "GSR_ICP_NN_KERNEL": "2"}, {"GSR_ICP_FUSED_STEP": "1"}, {"GSR_ICP_FUSED_STEP": "1", "GSR_ICP_NN_KERNEL": "2"},
                {"GSR_ICP_ADAPT": "1"}, {"GSR_ICP_ADAPT": "1", "GSR_ICP_CELL_TARGET": "64"}, {"GSR_ICP_ROBUST_BOX": "0"},
                {"GSR_ICP_PERSISTENT": "1"}, {"GSR_ICP_PERSISTENT": "1", "GSR_ICP_BLOCK_SEARCH": "0"}, {"GSR_ICP_PERSISTENT": "1", "GSR_ICP_BLOCKS": "100"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = run()
        for k in env:
            monkeypatch.delenv(k)
        for a, b in zip(ref[:2], got[:2]):
            assert np.abs(a[0] - b[0]).max() < 1e-12 and a[3] == b[3] and abs(a[1] - b[1]) < 1e-15 and abs(a[2] - b[2]) < 1e-12, env
        assert np.array_equal(ref[2][0], got[2][0]) and np.array_equal(ref[2][1], got[2][1]), env


def test_library_communicator_world1_rccl_and_side_stream():
    """The communicator of csrc/comm.hip on the box's one GPU: ncclCommInitRank with one rank (librccl opened lazily by the
    library), an all-reduce and an all-gather through it, and a registration whose per-iteration all-reduce the library
    enqueues itself (gsr_icp_set_comm): bit-identical to the plain registration.  The registration runs under a NON-default
    torch stream: the context's stream is that stream, and every launch and the collective are ordered on it."""
    from gaussiansplattingregistration_amd import icp, synth
    from gaussiansplattingregistration_amd.comm import Comm
    src, tgt, _ = synth.make_pair(40000, seed=17, sh_degree=0)
    nrm = icp.normals_from_cov(tgt["cov6"])
    with Comm.rccl(Comm.unique_id(), 0, 1, 0) as cm:
        assert cm.world == 1 and cm.transport == "rccl"
        t = torch.arange(32, dtype=torch.float64, device="cuda")
        cm.all_reduce(t)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(32, dtype=torch.float64))
        send = torch.arange(100, dtype=torch.uint8, device="cuda"); recv = torch.zeros(100, dtype=torch.uint8, device="cuda")
        cm.all_gather_bytes(send, recv)
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        side = torch.cuda.Stream()
        out = {}
        for tag in ("plain", "comm"):
            with torch.cuda.stream(side):
                with icp.IcpContext() as c:
                    c.set_target(tgt["xyz"], nrm, 0.3)
                    c.set_source(src["xyz"])
                    if tag == "comm":
                        c.set_comm(cm, src["xyz"].shape[0])
                    out[tag] = c.register(np.eye(4), kind=1, max_iter=25)
        a, b = out["plain"], out["comm"]
        assert np.array_equal(a["transformation"], b["transformation"]) and a["iterations"] == b["iterations"] > 3
        assert a["fitness"] == b["fitness"] and a["inlier_rmse"] == b["inlier_rmse"]


@pytest.mark.parametrize("max_corr", [0.08, 0.3, 1.5])
def test_far_outliers_keep_the_grid_robust_and_the_search_exact(oracle, monkeypatch, max_corr):
    """Scenes come with floaters: a dozen target points at 50 scene radii made the box of ALL points 10^5 times the scene's volume and
    the grid's cells swallowed the scene (bench.py --workload clustered: 116 s per registration).  The grid now lies over the box
    that holds all but 0.1 % of the points per side; the points beyond are clamped into the boundary cells, which the search treats
    as half-infinite.  The correspondences stay EXACT -- equal to the oracle's KD-tree and to the search over the raw box
    (GSR_ICP_ROBUST_BOX=0) -- also for queries far outside, for outliers that are each other's neighbours, and for targets sitting
    just outside the trimmed box; the registration equals the oracle's."""
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, T_gt = synth.make_pair(60000, seed=11, sh_degree=0, angle_deg=1.0)
    rng = np.random.default_rng(4)
    h = tgt["h"]
    tx, sx = tgt["xyz"].copy(), src["xyz"].copy()
    far = (rng.uniform(20.0, 60.0, (12, 1)) * h * rng.choice([-1.0, 1.0], (12, 3))).astype(np.float32)
    tx[:12] = far                                               # floaters in the target ...
    sx[100:106] = far[:6] + rng.normal(0, 0.02, (6, 3)).astype(np.float32)      # ... some of them with a source point right beside them
    sx[200:203] = far[6:9] * 1.5                                # ... and source points far from everything
    tx[300:340] = (tx[300:340] * 1.08).astype(np.float32)       # targets just outside the trimmed box
    T = np.eye(4); T[:3, 3] = 0.01
    res = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("GSR_ICP_ROBUST_BOX", knob)
        with icp.IcpContext() as c:
            c.set_target(tx, None, max_corr)
            c.set_source(sx)
            res[knob] = c.correspondences(T)
    widx, wd2 = oracle.icp_correspond(sx, tx, T, max_corr)
    for knob in ("1", "0"):
        assert np.array_equal(res[knob][0], widx), (knob, int((res[knob][0] != widx).sum()))
        assert np.allclose(res[knob][1], wd2, rtol=1e-12, atol=0)
    assert (widx[100:106] >= 0).all() and (widx[100:106] < 12).all()            # the floaters' companions found them
    monkeypatch.setenv("GSR_ICP_ROBUST_BOX", "1")
    nrm = icp.normals_from_cov(tgt["cov6"])
    r = icp.registration_icp_arrays(sx, tx, nrm, np.eye(4), kind=1, max_corr=max_corr, max_iter=15)
    w = oracle.icp(sx, tx, nrm, np.eye(4), kind=1, max_corr=max_corr, max_iter=15)
    assert r["iterations"] == w["iterations"] and np.linalg.norm(r["transformation"] - w["transformation"]) < 1e-5
    assert abs(r["fitness"] - w["fitness"]) < 1e-9


def test_registration_time_does_not_explode_with_floaters():
    """The symptom itself: 400 k points + 12 floaters at 20-60 h register in milliseconds, not seconds."""
    import time
    from gaussiansplattingregistration_amd import icp, synth
    src, tgt, _ = synth.make_pair(400000, seed=12, sh_degree=0, angle_deg=1.0)
    rng = np.random.default_rng(5)
    far = (rng.uniform(20.0, 60.0, (12, 1)) * tgt["h"] * rng.choice([-1.0, 1.0], (12, 3))).astype(np.float32)
    tx, sx = tgt["xyz"].copy(), src["xyz"].copy()
    tx[:12] = far
    sx[50:62] = far * 0.9
    nrm = icp.normals_from_cov(tgt["cov6"])
    icp.registration_icp_arrays(sx, tx, nrm, np.eye(4), kind=1, max_corr=0.2, max_iter=5)      # warm-up (allocation)
    t = time.perf_counter()
    r = icp.registration_icp_arrays(sx, tx, nrm, np.eye(4), kind=1, max_corr=0.2, max_iter=10)
    dt = time.perf_counter() - t
    assert r["fitness"] > 0.9 and dt < 0.5, (r["fitness"], dt)
