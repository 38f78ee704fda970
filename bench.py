#!/usr/bin/env python3
"""Benchmark of the hot path: HEM mixture levels + coarse-to-fine ICP on a synthetic splat pair.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n SPLATS] [--no-cpu-baseline]

One STEP = one pass of the whole hot path over one pair of clouds that is already resident in HBM:
  3 HEM levels on the source cloud and on the target cloud (rho=3, delta=3, kappa=2.5, tau=1 --
  src/params/merge_parameters.py:5-10), covariance normals for every level, then the 4-entry
  coarse-to-fine point-to-plane ICP (iter_values [50,30,20,10], max_corr [0.5,0.3,0.2,0.1]).
Workload at N=1 = BASELINE.json configs[2] (the configuration the metric is quoted on): 2 x 5M splats,
SH degree 3.  With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
runs its own pair -- the path shards by independent clouds with no data-path collective (weak
scaling); timing is barrier-bracketed and the MAX over ranks is taken.

Prints ONE JSON line on rank 0.  `value` = level-input Gaussians per second through the HEM levels
(whole job); `icp_iters_per_sec` rides along; `roofline` is for the dominant kernel of the step,
`cpu_baseline` is the reference's own compiled extension (oracle/_ref) -- or the oracle port when
that is absent -- timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEM_PARAMS = dict(hem_reduction=3.0, distance_delta=3.0, color_delta=2.5, decay_rate=1.0)
LEVELS = 3
ITER_VALUES = [50, 30, 20, 10]
MAX_CORR = [0.5, 0.3, 0.2, 0.1]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy rate)
B_GEOM = 57                    # bytes/component the selection kernel must read once: xyz 12 color 12 cov 24 opacity 4 weight 4 flag 1


def bytes_level(n_in, n_out, F):
    """algorithmic bytes of one HEM level."""
    B = 57 + 4 * F
    return n_in * (B + 16) + n_out * B


def hot_path_step(ctxs, lru, PointCloud, src, tgt, device, sync):
    """One pass over one pair.  `ctxs` holds the long-lived library contexts (their workspaces are
    reused from step to step: no allocation in steady state)."""
    out = {"hem_gaussians": 0, "hem_s": 0.0, "icp_s": 0.0, "icp_iters": 0, "levels": [], "kern": []}
    clouds = []
    t0 = time.perf_counter()
    m = ctxs["hem"]
    m.set_rng("glibc", 1, 0)                       # a fresh reference process: cloud 1 then cloud 2 on one stream
    for c in (src, tgt):
        lv = [PointCloud(xyz32=c["xyz"], cov6=c["cov6"])]
        m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)     # inputs resident in HBM: read in place
        for _ in range(LEVELS):
            m.run_level()
            st = m.stats()
            out["hem_gaussians"] += st["n_in"]
            out["kern"].append(st)
            d = m.get_level(as_torch=True)
            lv.append(PointCloud(xyz32=d["xyz"], cov6=d["cov6"]))
        clouds.append(lv)
    sync()
    t1 = time.perf_counter()
    out["hem_s"] = t1 - t0
    out["levels"] = [len(p) for p in clouds[0]]
    # coarse-to-fine ICP over the level lists (qt_multiscale_registrator.py:197-236)
    T = np.eye(4)
    est = lru.get_estimation(lru.LocalRegistrationType.ICP_Point_To_Plane, lru.RobustLoss(0))
    icp_kernel_ms, icp_kernels = 0.0, 0
    for k in range(LEVELS + 1):
        s, t = clouds[0][-(k + 1)], clouds[1][-(k + 1)]
        t.estimate_normals()
        crit = lru.get_convergence_criteria(1e-6, 1e-6, ITER_VALUES[k])
        r = lru.registration_icp(s, t, MAX_CORR[k], T, est, crit, device=device, ctx=ctxs["icp"])
        T = r.transformation
        out["icp_iters"] += r.iterations
        icp_kernel_ms += r.timing["ms_iters"]
        icp_kernels += r.timing["iter_kernels"]
        out["icp_finest"] = {"ns": len(s), "ms": r.timing["ms_iters"], "kernels": r.timing["iter_kernels"]}
    sync()
    out["icp_s"] = time.perf_counter() - t1
    out["T"] = T
    out["fitness"], out["rmse"] = r.fitness, r.inlier_rmse
    out["icp_kernel_ms"], out["icp_kernels"] = icp_kernel_ms, icp_kernels
    return out


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/*_pmc.json,
    written by scripts/summarize_profiles.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    passes of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=os.path.basename)
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "hbm_read_bytes_per_launch_x2_corrected" in v:
                return v["hbm_read_bytes_per_launch_x2_corrected"] + v.get("hbm_write_bytes_per_launch", 0.0), os.path.basename(f)
    return None, None


def pmc_valu(kernel_prefix):
    """VALU instructions per launch of the dominant kernel and the share of the launch they occupy (SQ_INSTS_VALU x 4
    cycles / 1024 SIMDs against SQ_WAVE-independent wall cycles at 2.4 GHz), from the newest committed PMC summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), key=os.path.basename)
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "SQ_INSTS_VALU_per_launch" in v:
                return float(v["SQ_INSTS_VALU_per_launch"]), os.path.basename(f)
    return None, None


def cpu_baseline():
    """The reference's own extension (or the oracle port) on this box's host cores, bounded sample."""
    from gaussiansplattingregistration_amd import synth
    cores = os.cpu_count() or 1
    n = 500000                     # ~10-20 s on the host cores (the reference slows down per splat as clouds grow)
    cloud = synth.make_cloud(n, seed=0)
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    have_ref = os.path.isdir(ref_dir) and any(f.startswith("mixture_bind") for f in os.listdir(ref_dir))
    res = {"unit": "Gaussians/s", "cores": cores}
    if have_ref:
        try:
            with tempfile.TemporaryDirectory() as td:
                inp, outp = os.path.join(td, "i.npz"), os.path.join(td, "o.npz")
                np.savez(inp, levels=1, rho=3.0, delta=3.0, kappa=2.5, tau=1.0,
                         **{k: cloud[k] for k in ("xyz", "color", "opacity", "cov6", "sh")})
                subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_ref.py"), inp, outp, "--threads", str(cores)],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                o = np.load(outp)
                wall = float(o["wall_s"])
            res.update(value=n / wall, kind="reference",
                       sample=f"reference cpp_ext (oracle/_ref) CreateMixture, 1 level, {n} splats SH deg 3 at bench density, "
                              f"{cores} OpenMP threads, list marshalling excluded; wall {wall:.2f} s")
        except Exception as e:  # pragma: no cover
            have_ref = False
            res["ref_error"] = str(e)[:200]
    from oracle import oracle as O
    if not have_ref:
        t = time.perf_counter()
        O.hem(cloud, 1, threads=cores)
        wall = time.perf_counter() - t
        res.update(value=n / wall, kind="port",
                   sample=f"oracle port (oracle/hem_oracle.cpp), 1 level, {n} splats SH deg 3, {cores} OpenMP threads; wall {wall:.2f} s")
    # ICP: the oracle port (Open3D absent) -- KD-tree ICP, point-to-plane
    ns = 200000
    src, tgt, _ = synth.make_pair(ns, seed=1, sh_degree=0, angle_deg=1.0)
    C = tgt["cov6"].astype(np.float64)
    nrm = O.normals_from_cov(np.stack([C[:, [0, 1, 2]], C[:, [1, 3, 4]], C[:, [2, 4, 5]]], 1))
    t = time.perf_counter()
    r = O.icp(src["xyz"], tgt["xyz"], nrm, np.eye(4), kind=1, max_corr=0.1, max_iter=5, rel_fitness=0, rel_rmse=0, threads=cores)
    wall = time.perf_counter() - t
    res.update(icp_iters_per_sec=r["iterations"] / wall, icp_kind="port",
               icp_sample=f"oracle KD-tree ICP (point-to-plane), {ns} source x {ns} target points, 5 iterations, {cores} threads; wall {wall:.2f} s "
                          "(tree build included)")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--splats", "--n", dest="n", type=int, default=5_000_000, help="splats per cloud (use --splats under torchrun: its parser claims --n)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import __graft_entry__ as g
    from gaussiansplattingregistration_amd import parallel
    rank, world, local_rank = parallel.init_distributed()
    if rank == 0:
        g.build_hip()                       # one rank (re)builds a stale library, the others wait for it
    if world > 1:
        torch.distributed.barrier()
    from gaussiansplattingregistration_amd import hem, icp as icp_mod, synth
    from gaussiansplattingregistration_amd.models.point_cloud import PointCloud
    from gaussiansplattingregistration_amd.utils import local_registration_util as lru
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: this backend has no CPU fallback")
    device = local_rank if world > 1 else 0
    if os.environ.get("GSR_BENCH_SAME_DEVICE"):          # test hook: several ranks on one GPU (with GSR_DIST_BACKEND=gloo)
        device = 0
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)

    def sync():
        torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # synthetic pair, resident in HBM before the timed region: target = cloud, source = inv(T_gt) * cloud + jitter
    n = a.n
    tgt = synth.make_cloud_torch(n, seed=100 + rank, device=dev)
    T_gt = synth.rigid_transform(1.0, (1, 1, 1), 0.004 * tgt["h"] * np.array([1.0, -1.0, 0.5]))
    src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    src["xyz"] = src["xyz"] + torch.randn(src["xyz"].shape, device=dev, generator=gen) * 0.002
    src = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in src.items()}
    sync()

    ctxs = {"hem": hem.HemMixture(device=device, rng_mode="glibc", **HEM_PARAMS), "icp": icp_mod.IcpContext(device=device)}
    for _ in range(a.warmup):
        hot_path_step(ctxs, lru, PointCloud, src, tgt, device, sync)
    sync(); barrier()
    t0 = time.perf_counter()
    runs = [hot_path_step(ctxs, lru, PointCloud, src, tgt, device, sync) for _ in range(a.steps)]
    sync(); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        rdev = dev if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([sum(r["hem_s"] for r in runs), sum(r["icp_s"] for r in runs)], dtype=torch.float64, device=rdev)
        torch.distributed.all_reduce(agg, op=torch.distributed.ReduceOp.MAX)
        hem_s, icp_s = float(agg[0]), float(agg[1])
    else:
        hem_s, icp_s = sum(r["hem_s"] for r in runs), sum(r["icp_s"] for r in runs)

    # measured device-copy bandwidth of this very GPU (SURVEY 8d asks for the fraction of both the nominal and a measured
    # figure): 1 GiB device-to-device, read + write traffic
    copy_gbs = None
    if rank == 0:
        try:
            src_t = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            dst_t = torch.empty_like(src_t)
            dst_t.copy_(src_t); sync()
            tc = time.perf_counter()
            for _ in range(5):
                dst_t.copy_(src_t)
            sync()
            copy_gbs = 5 * 2.0 * src_t.numel() * 4 / (time.perf_counter() - tc) / 1e9
            del src_t, dst_t
        except Exception:
            copy_gbs = None

    if rank == 0:
        hem_gauss = sum(r["hem_gaussians"] for r in runs) * world
        icp_iters = sum(r["icp_iters"] for r in runs) * world
        F = 45
        # dominant kernel of the step: k_select<FILL> (selection + likelihood), timed by hipEvent pairs on the
        # library's stream around every launch (gsr_hem_get_phase_ms[7])
        kern = [k for r in runs for k in r["kern"]]
        phases = {p: sum(k[p] for k in kern) for p in ("ms_grid", "ms_select", "ms_sumlw", "ms_mstep", "ms_flags", "ms_level",
                                                         "ms_k_select_count", "ms_k_select_fill")}
        fill_ms = np.array([k["ms_k_select_fill"] for k in kern])
        n_in = np.array([k["n_in"] for k in kern], dtype=np.float64)
        avg_ms = float(fill_ms.mean())
        achieved = float(n_in.mean()) * B_GEOM / (avg_ms * 1e-3) / 1e9
        lvl1 = [k for k in kern if k["n_in"] == n]
        lvl_bytes = np.mean([bytes_level(k["n_in"], k["n_out"], F) for k in lvl1])
        lvl_ms = np.mean([k["ms_level"] for k in lvl1])
        traffic, traffic_src = pmc_traffic("gsr::k_select<2")
        line = {
            "metric": "Gaussians/sec through HEM level + ICP iters/sec, 2x5M-splat pair",
            "value": hem_gauss / hem_s,
            "unit": "Gaussians/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (HEM) / f64 (ICP)", "data": "synthetic",
            "config": {"workload": f"2x{n} synthetic splats (SH deg 3) per GPU: 3 HEM levels per cloud + 4-level coarse-to-fine "
                                   "point-to-plane ICP (BASELINE configs[2])",
                       "hem_params": HEM_PARAMS, "iter_values": ITER_VALUES, "max_corr": MAX_CORR, "level_sizes": runs[-1]["levels"],
                       "parallelism": f"{world} independent pair(s), one per GPU, no data-path collective"},
            "icp_iters_per_sec": icp_iters / icp_s,
            "icp_iterations_per_step": runs[-1]["icp_iters"],
            "hem_s_per_step": hem_s / a.steps, "icp_s_per_step": icp_s / a.steps,
            "icp_result": {"fitness": runs[-1]["fitness"], "inlier_rmse": runs[-1]["rmse"],
                           "T_err_vs_ground_truth_F": float(np.linalg.norm(runs[-1]["T"] - T_gt))},
            "hem_phase_ms_per_step": {k: v / a.steps for k, v in phases.items()},
            "roofline": {"bound": "hbm", "kernel": "k_select<SPARSE> (child selection + likelihood, one wavefront per parent)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "measured_copy_GBps": copy_gbs, "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "avg_launch_ms": avg_ms, "launches": int(len(fill_ms)), "avg_units_per_launch": float(n_in.mean()),
                         "bytes_per_unit": B_GEOM,
                         "valu": (lambda vi: None if vi[0] is None else {
                             "insts_per_launch": vi[0], "source": vi[1],
                             "busy_frac_at_2.4GHz": vi[0] * 4.0 / 1024.0 / (avg_ms * 1e-3 * 2.4e9)})(pmc_valu("gsr::k_select<2")),
                         "note": "VALU-bound neighbour evaluation (about 240 candidate tests and 80 KL divergences per splat; PMC: VALUBusy 94 %), not HBM-bound: see DESIGN.md section 4",
                         "level1": {"algorithmic_bytes": float(lvl_bytes), "ms": float(lvl_ms),
                                    "achieved_GBps": float(lvl_bytes / (lvl_ms * 1e-3) / 1e9),
                                    "frac": float(lvl_bytes / (lvl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)},
                         "icp_finest": {"ns": runs[-1]["icp_finest"]["ns"],
                                        "avg_ms": runs[-1]["icp_finest"]["ms"] / max(1, runs[-1]["icp_finest"]["kernels"]),
                                        "achieved_GBps": 48.0 * runs[-1]["icp_finest"]["ns"] /
                                        (runs[-1]["icp_finest"]["ms"] / max(1, runs[-1]["icp_finest"]["kernels"]) * 1e-3) / 1e9}},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
