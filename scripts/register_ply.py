#!/usr/bin/env python3
"""Headless registration of two 3DGS .ply scenes on one MI355X: what the reference's GUI does through its "Mixture" and
"Multiscale registration" tabs (qt_gaussian_mixture.py, qt_multiscale_registrator.py), as a script.

    python scripts/register_ply.py first.ply second.ply --levels 3 --max-corr 0.5 0.3 0.2 0.1 --iters 50 30 20 10 \\
           [--type plane|point|color|general] [--loss none|tukey|cauchy|gm|huber --k 0.1] [--voxel] [--out merged.ply]

Prints the 4x4 transformation (first -> second), fitness and inlier RMSE; `--out` saves the merged cloud.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("first")
    ap.add_argument("second")
    ap.add_argument("--levels", type=int, default=3, help="HEM mixture levels per cloud (ignored with --voxel)")
    ap.add_argument("--max-corr", type=float, nargs="+", default=[0.5, 0.3, 0.2, 0.1], help="coarse -> fine (voxel sizes with --voxel)")
    ap.add_argument("--iters", type=int, nargs="+", default=[50, 30, 20, 10])
    ap.add_argument("--type", choices=["point", "plane", "color", "general"], default="plane")
    ap.add_argument("--loss", choices=["none", "tukey", "cauchy", "gm", "huber"], default="none")
    ap.add_argument("--k", type=float, default=0.0)
    ap.add_argument("--voxel", action="store_true", help="voxel multiscale path instead of HEM mixtures")
    ap.add_argument("--hem", type=float, nargs=4, default=[3.0, 3.0, 2.5, 1.0], metavar=("RHO", "DELTA", "KAPPA", "TAU"))
    ap.add_argument("--out")
    a = ap.parse_args()

    import __graft_entry__ as g
    g.build_hip()
    from gaussiansplattingregistration_amd import mixture_bind
    from gaussiansplattingregistration_amd.controllers.downsampler_controller import DownsamplerController
    from gaussiansplattingregistration_amd.controllers.registration_controller import RegistrationController
    from gaussiansplattingregistration_amd.models.data_repository import DataRepository, UIStateRepository
    from gaussiansplattingregistration_amd.models.gaussian_model import GaussianModel
    from gaussiansplattingregistration_amd.params import GaussianMixtureParams
    from gaussiansplattingregistration_amd.utils.local_registration_util import KernelLossFunctionType as K, LocalRegistrationType as T
    from gaussiansplattingregistration_amd.utils.point_cloud_converter import convert_gs_to_open3d_pc

    rtype = {"point": T.ICP_Point_To_Point, "plane": T.ICP_Point_To_Plane, "color": T.ICP_Color, "general": T.ICP_General}[a.type]
    loss = {"none": K.Loss_None, "tukey": K.Tukey_Loss, "cauchy": K.Cauchy_Loss, "gm": K.GMLoss, "huber": K.Huber_Loss}[a.loss]
    repo, ui = DataRepository(), UIStateRepository()
    t0 = time.perf_counter()
    for path, gl, ol in ((a.first, repo.pc_gaussian_list_first, repo.pc_open3d_list_first),
                         (a.second, repo.pc_gaussian_list_second, repo.pc_open3d_list_second)):
        tm = {}
        gm = GaussianModel("cuda:0").from_ply(path, timing=tm)        # pinned chunks -> HBM -> device SoA (utils/ply_io.load_gaussian_device)
        gl.append(gm)
        ol.append(convert_gs_to_open3d_pc(gm))
        print(f"{path}: {len(gm)} splats, SH degree {gm.sh_degree}; file -> device arrays {tm['seconds'] * 1e3:.1f} ms "
              f"({tm['bytes'] / tm['seconds'] / 1e9:.2f} GB/s of file bytes)")
    t1 = time.perf_counter()
    if not a.voxel:
        if len(a.max_corr) != a.levels + 1 or len(a.iters) != a.levels + 1:
            raise SystemExit("--max-corr and --iters need levels + 1 values (coarsest first)")
        mixture_bind.reset_rng()
        rho, delta, kappa, tau = a.hem
        DownsamplerController(repo).create_mixture(GaussianMixtureParams(hem_reduction=rho, distance_delta=delta, color_delta=kappa,
                                                                         decay_rate=tau, cluster_level=a.levels))
        print("levels:", [len(x) for x in repo.pc_gaussian_list_first], "/", [len(x) for x in repo.pc_gaussian_list_second])
    t2 = time.perf_counter()
    rc = RegistrationController(repo, ui)
    res = rc.execute_multiscale_registration(False, "", "", rtype, 1e-6, 1e-6, a.max_corr, a.iters, loss, a.k, not a.voxel)
    t3 = time.perf_counter()
    if res is None:
        raise SystemExit("registration failed: " + "; ".join(rc.errors))
    np.set_printoptions(precision=6, suppress=True)
    print("transformation (first -> second):\n", res.result.transformation)
    print(f"fitness {res.result.fitness:.4f}  inlier RMSE {res.result.inlier_rmse:.6f}")
    print(f"load {t1 - t0:.2f} s, mixtures {t2 - t1:.3f} s, registration {t3 - t2:.3f} s")
    if a.out:
        merged = GaussianModel.get_merged_gaussian_point_clouds(repo.pc_gaussian_list_first[0], repo.pc_gaussian_list_second[0],
                                                                res.result.transformation)
        merged.save_ply(a.out)
        print(f"merged cloud ({len(merged)} splats) -> {a.out}")


if __name__ == "__main__":
    main()
