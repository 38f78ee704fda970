"""The sparse pre-registration branch of the multiscale worker (reference qt_multiscale_registrator.py:45-46,74-90,
file_loader.py:20-30, point_cloud_converter.py:9-28): sparse input clouds (x y z red green blue .ply) -> KNN-30 normals
-> one ICP whose transformation seeds the coarse-to-fine loop over the mixture levels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sheet(n, seed, T=None):
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1, 1, (n, 2))
    z = 0.25 * np.sin(3.0 * u[:, 0]) * np.cos(2.0 * u[:, 1]) + rng.normal(0, 0.004, n)
    xyz = np.stack([u[:, 0], u[:, 1], z], 1)
    xyz[: n // 6] = rng.normal(0, 0.15, (n // 6, 3)) + np.array([0.3, -0.2, 0.6])
    if T is not None:
        xyz = xyz @ T[:3, :3].T + T[:3, 3]
    return xyz.astype(np.float32)


def test_knn_normals_equal_the_oracle(oracle):
    """gsr_normals_knn against the oracle's restatement of Open3D's EstimateNormals(KNN 30): same neighbours, same
    cumulant covariance, same FastEigen3x3 -> 1e-9; plus the degenerate inputs (duplicates, fewer than 3 points)."""
    from gaussiansplattingregistration_amd import icp
    xyz = _sheet(20000, 3)
    xyz[100:110] = xyz[90:100]                                    # exact duplicates
    xyz[200:203] += np.float32([50.0, 0.0, 0.0])                  # a far-away triple
    got = icp.normals_knn(xyz, 30)
    want = oracle.normals_knn(xyz.astype(np.float64), 30)
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-12
    assert np.abs(got - want).max() < 1e-9
    for knn in (3, 7):
        assert np.abs(icp.normals_knn(xyz[:5000], knn) - oracle.normals_knn(xyz[:5000].astype(np.float64), knn)).max() < 1e-9
    two = np.float32([[0, 0, 0], [1, 0, 0]])
    assert np.array_equal(icp.normals_knn(two, 30), [[0, 0, 1], [0, 0, 1]])         # < 3 neighbours: identity covariance -> (0, 0, 1)
    import torch
    dev = icp.normals_knn(torch.from_numpy(xyz).cuda(), 30)
    assert np.array_equal(dev.cpu().numpy(), got)


def test_multiscale_with_sparse_pre_registration(tmp_path, oracle):
    from gaussiansplattingregistration_amd import hem, synth
    from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
    from gaussiansplattingregistration_amd.utils import file_loader, ply_io
    from gaussiansplattingregistration_amd.utils.local_registration_util import KernelLossFunctionType, LocalRegistrationType, do_icp_registration
    from gaussiansplattingregistration_amd.workers.registrators import MultiScaleRegistratorMixture
    T_gt = synth.rigid_transform(6.0, (0.3, 1.0, 0.2), (0.06, -0.04, 0.03))
    # sparse input clouds of the two scenes (what COLMAP leaves next to a 3DGS training run)
    rng = np.random.default_rng(0)
    sa, sb = _sheet(6000, 11), _sheet(6000, 11, T_gt)
    pa, pb = tmp_path / "sparse_a.ply", tmp_path / "sparse_b.ply"
    ply_io.save_input_ply(pa, sa, rng.integers(0, 256, (6000, 3)))
    ply_io.save_input_ply(pb, sb, rng.integers(0, 256, (6000, 3)), normals=np.zeros((6000, 3)))
    assert file_loader.check_point_cloud_type(ply_io.read_ply_vertices(pa)) is file_loader.PointCloudType.INPUT
    spc = file_loader.load_sparse_pc(str(pa))
    assert len(spc) == 6000 and spc.has_normals() and spc.cov6 is None and float(np.max(spc.colors)) <= 1.0
    assert np.abs(spc.normals - oracle.normals_knn(sa.astype(np.float64), 30)).max() < 1e-9
    assert file_loader.load_sparse_pc(str(tmp_path / "missing.ply")) is None
    # dense splat clouds of the same two scenes and their mixture levels
    def levels(xyz_seed, T):
        c = synth.make_cloud(30000, seed=5, h=1.0, sh_degree=0)
        c["xyz"] = _sheet(30000, xyz_seed, T)
        c["cov6"] = (c["cov6"] * np.float32(0.02)).astype(np.float32)
        lv, _ = hem.create_mixture(c, 2)
        out = [PointCloud(xyz32=c["xyz"], cov6=c["cov6"]).estimate_normals()]
        return out + [PointCloud(xyz32=l["xyz"], cov6=l["cov6"]).estimate_normals() for l in lv]
    l1, l2 = levels(21, None), levels(21, T_gt)
    vox, its = [0.3, 0.15, 0.08], [40, 30, 20]
    kind = LocalRegistrationType.ICP_Point_To_Plane
    w = MultiScaleRegistratorMixture(l1, l2, np.eye(4), True, str(pa), str(pb), kind, 1e-7, 1e-7, vox, its, KernelLossFunctionType.Loss_None, 0.0)
    out = w.run()
    assert out is not None, w.errors
    # = the sparse ICP first, then the loop seeded with its transformation (qt_multiscale_registrator.py:45-46)
    spc2 = file_loader.load_sparse_pc(str(pb))
    r0 = do_icp_registration(spc, spc2, np.eye(4), kind, vox[0], 1e-7, 1e-7, its[0], KernelLossFunctionType.Loss_None, 0.0)
    assert np.array_equal(w.sparse_result.transformation, r0.transformation)
    assert np.linalg.norm(r0.transformation - T_gt) < 0.05
    w2 = MultiScaleRegistratorMixture(l1, l2, r0.transformation, False, "", "", kind, 1e-7, 1e-7, vox, its, KernelLossFunctionType.Loss_None, 0.0)
    out2 = w2.run()
    assert np.array_equal(out.result.transformation, out2.result.transformation)
    assert np.linalg.norm(out.result.transformation - T_gt) < 2e-2
    assert out.registration_data.used_sparse_clouds is True
    # a Gaussian .ply offered as a sparse cloud is refused with the reference's message
    ply_io.save_gaussian_ply(tmp_path / "g.ply", sa, np.zeros((6000, 3)), np.zeros((6000, 0)), np.zeros(6000), np.zeros((6000, 3)),
                             np.tile(np.float32([1, 0, 0, 0]), (6000, 1)))
    w3 = MultiScaleRegistratorMixture(l1, l2, np.eye(4), True, str(tmp_path / "g.ply"), str(pb), kind, 1e-7, 1e-7, vox, its,
                                      KernelLossFunctionType.Loss_None, 0.0)
    assert w3.run() is None and w3.errors == ["Point clouds provided as sparse were of a different type"]
