#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] at their workload size on ONE GPU, the ranks sharing it.

    python scripts/fullsize_modes.py c5 [--splats 40000000] [--world 8] [--levels 3] [--transport mock|gloo] [--out file.json]
    python scripts/fullsize_modes.py c4 [--splats 5000000] [--transport mock|gloo] [--out file.json]

c5  (SURVEY 8(e) row 3).  The cloud is synth.make_block_cloud_torch's: defined block by block, global indices contiguous per block.
    1. ONE process, ONE context: the `world` blocks concatenated, `levels` HEM levels; per level and array a 64-bit hash of the rows
       keyed by their row number (= global index).
    2. `world` fresh processes on device 0, each with ITS block only: gsr_hem_set_level0_part + partitioned levels through the
       library's communicator -- the RCCL branch of csrc/comm.hip over tests/mock_rccl (`--transport mock`, default) or the callback
       transport over gloo.  Every rank hashes its piece keyed by the rows' global indices; the hashes add up (mod 2^64).
    3. Equal hashes for every array of every level = the assembled partitioned levels ARE the single-context levels, bit for bit.
    Recorded per rank and level: owned rows, ghosts, rows sent, halo bytes (72-byte rows and SH rows apart), bytes of the four sum
    exchanges, bit-map bytes, heavy parents / bucket overflow / one-pass flags, level time.

c4  (SURVEY 8(e) rows 1 + 2).  bench.py's own pair.
    1. ONE process: the levels of cloud A and of cloud B (cloud k from rand() position k * bench.C4_STREAM_STRIDE: two independent clouds), then the 4-entry
       coarse-to-fine point-to-plane ICP.
    2. TWO fresh processes on device 0: cloud A's levels on rank 0, cloud B's on rank 1, one broadcast of the level lists, ICP with the
       source split over both ranks and the library's all-reduce of 32 float64 per iteration (bench.step_c4).
    3. The final transform must agree to 1e-9 (Frobenius), the iteration counts per level must be equal.

No process that has touched the GPU is ever replaced: the parent never initialises HIP, every phase is a fresh child (subprocess.Popen).
Exit code 0 iff the comparison holds.  No xGMI is involved: the ranks share one device, so there is NO collective TIME here, only bytes.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 100


# ------------------------------------------------------------------------------------------------ hashing (device, torch int64 wraps)
def _i64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


_M1, _M2, _M3 = _i64(0x9E3779B97F4A7C15), _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)


def rows_hash(t, gid, chunk=1 << 20):
    """Sum over the rows of mix(gid, the row's 32-bit words), mod 2^64: independent of how the rows are split over ranks."""
    import torch
    t = t.reshape(t.shape[0], -1)
    total = 0
    for lo in range(0, t.shape[0], chunk):
        w = t[lo:lo + chunk].contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        acc = gid[lo:lo + chunk].to(torch.int64) * _M1 + _M2
        for j in range(w.shape[1]):
            acc = (acc ^ w[:, j]) * _M3
            acc = acc ^ (acc >> 29)
        total = (total + int(acc.sum().item())) & ((1 << 64) - 1)
    return total


def level_hashes(d, gid):
    return {k: "%016x" % rows_hash(d[k], gid) for k in ("xyz", "color", "cov6", "opacity", "sh")}


# ------------------------------------------------------------------------------------------------ workers
def _setup_dist(a):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from gaussiansplattingregistration_amd.comm import Comm
    return Comm.from_torch_group(0)        # GSR_COMM_TRANSPORT=rccl + GSR_RCCL_LIB (the mock) -> the library's RCCL branch; else callbacks


def worker_c5_single(a):
    import torch
    import bench
    from gaussiansplattingregistration_amd import hem, synth
    dev = torch.device("cuda", 0)
    parts = [synth.make_block_cloud_torch(a.splats, r, a.blocks, seed=SEED, device=dev)[0] for r in range(a.blocks)]
    cloud = {k: torch.cat([p[k] for p in parts]).contiguous() for k in ("xyz", "color", "opacity", "cov6", "sh")}
    del parts
    torch.cuda.synchronize()
    res = {"n": a.splats, "levels": []}
    with hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS) as m:
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"], borrow=True)
        for _ in range(a.levels):
            t0 = time.perf_counter()
            m.run_level()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            st = m.stats()
            d = m.get_level(as_torch=True)
            gid = torch.arange(d["xyz"].shape[0], device=dev)
            res["levels"].append({"n_in": st["n_in"], "n_out": st["n_out"], "parents": st["parents"], "pairs": st["pairs"], "orphans": st["orphans"],
                                  "dropped": st["dropped"], "ms_level": st["ms_level"], "wall_ms": wall * 1e3, "heavy_parents": st["heavy_parents"],
                                  "one_pass": st["one_pass"], "partition_overflow": st["partition_overflow"], "hash": level_hashes(d, gid)})
            del d
    res["peak_device_GB"] = torch.cuda.max_memory_allocated() / 1e9
    json.dump(res, open(a.result, "w"))


def worker_c5_part(a):
    import torch
    import torch.distributed as dist
    import bench
    from gaussiansplattingregistration_amd import hem, parallel, synth
    comm = _setup_dist(a)
    dev = torch.device("cuda", 0)
    blk, gid0 = synth.make_block_cloud_torch(a.splats, a.rank, a.world, seed=SEED, device=dev)
    torch.cuda.synchronize()
    m = hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS)
    m.set_timing(2)
    dist.barrier()
    t0 = time.perf_counter()
    pieces, st = parallel.hem_partitioned(blk, a.levels, comm, device=0, as_torch=True, owned=gid0, n_global=a.splats, mixture=m, rng_mode="glibc")
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    res = {"rank": a.rank, "transport": comm.transport, "wall_s_all_levels": wall, "levels": []}
    F = int(blk["sh"].shape[1])
    n_glob_in = a.splats
    for p, s in zip(pieces, st):
        gid = torch.from_numpy(p["gid"].astype("int64")).to(dev)
        cells = int(s["cells"])
        row = {"owned_in": s["n_in"], "owned_out": s["n_out"], "ghosts": s["ghosts"], "rows_sent": s["rows_sent"],
               "halo_bytes_received": s["halo_bytes_received"], "halo_row_bytes_received": s["ghosts"] * 72, "halo_sh_bytes_received": s["ghosts"] * 4 * F,
               "sum_exchange_bytes_received": s["sum_exchange_bytes_received"],
               # two bit maps over the level's global indices, summed over the ranks (parents, orphans), + one more when a row is erased
               "bitmap_bytes_allreduced": 2 * ((n_glob_in + 31) // 32 + 1) * 4,
               "cell_mask_bytes_allgathered_per_rank": 2 * ((cells + 31) // 32 + 1) * 4,
               "parents": s["parents"], "pairs": s["pairs"], "candidates": s["candidates"], "heavy_parents": s["heavy_parents"],
               "heavy_work_items": s["heavy_work_items"], "one_pass": s["one_pass"], "partition_overflow": s["partition_overflow"],
               "max_pairs_of_a_parent": s["max_pairs_of_a_parent"], "dropped_global": s["dropped_global"], "n_global_out": s["n_global"],
               "ms_level": s["ms_level"], "ms_halo_rows": s["ms_halo_rows"], "ms_halo_sh": s["ms_halo_sh_overlapped"],
               "ms_k_select": s["ms_k_select"], "ms_k_mstep": s["ms_k_mstep"], "hash": level_hashes(p, gid), "rows": int(gid.numel())}
        n_glob_in = int(s["n_global"])
        res["levels"].append(row)
    res["peak_device_GB"] = torch.cuda.max_memory_allocated() / 1e9
    json.dump(res, open(a.result, "w"))
    dist.barrier()
    m.close()
    comm.close()
    dist.destroy_process_group()


def _c4_pair(n, dev):
    """bench.py's pair (main()): target = cloud of seed 100, source = inv(T_gt) * target + jitter (generator seed 7)."""
    import numpy as np
    import torch
    import bench
    from gaussiansplattingregistration_amd import synth
    tgt = synth.make_cloud_torch(n, seed=SEED, device=dev, shape="iso")
    T_gt = synth.rigid_transform(bench.PAIR_ANGLE_DEG, (1, 1, 1), bench.PAIR_SHIFT_H * tgt["h"] * np.array([1.0, -1.0, 0.5]))
    src = synth.apply_rigid_torch(tgt, np.linalg.inv(T_gt))
    gen = torch.Generator(device=dev).manual_seed(7)
    src["xyz"] = src["xyz"] + torch.randn(src["xyz"].shape, device=dev, generator=gen) * 0.002
    src = {k: (v.contiguous() if isinstance(v, torch.Tensor) else v) for k, v in src.items()}
    torch.cuda.synchronize()
    return src, tgt, T_gt


def _icp_record(out, T_gt):
    import numpy as np
    return {"T": [[float(v) for v in r] for r in out["T"]], "fitness": float(out["fitness"]), "rmse": float(out["rmse"]),
            "iterations": [l["iterations"] for l in out["levels"]], "ns": [l["ns"] for l in out["levels"]], "nt": [l["nt"] for l in out["levels"]],
            "ms_per_iteration": [l["ms_iters"] / max(1, l["evals"]) for l in out["levels"]],
            "T_err_vs_ground_truth_F": float(np.linalg.norm(np.asarray(out["T"]) - T_gt))}


def worker_c4_single(a):
    import torch
    import bench
    from gaussiansplattingregistration_amd import hem, icp as icp_mod
    from gaussiansplattingregistration_amd.utils import local_registration_util as lru
    dev = torch.device("cuda", 0)
    src, tgt, T_gt = _c4_pair(a.splats, dev)
    m = hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS)
    lists = []
    for ci, c in enumerate((src, tgt)):
        m.set_rng("glibc", 1, ci * bench.C4_STREAM_STRIDE)     # every cloud from its own stream position: two independent clouds (step_c4 does the same)
        lists.append(bench.hem_levels(m, c)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = bench.coarse_to_fine(lru, icp_mod.IcpContext(device=0), lists[0], lists[1], 0)
    torch.cuda.synchronize()
    res = _icp_record(out, T_gt)
    res.update(level_sizes=[len(p) for p in lists[0]], icp_s=time.perf_counter() - t0)
    json.dump(res, open(a.result, "w"))


def worker_c4_part(a):
    import torch
    import torch.distributed as dist
    import bench
    from gaussiansplattingregistration_amd import hem, icp as icp_mod
    from gaussiansplattingregistration_amd.utils import local_registration_util as lru
    comm = _setup_dist(a)
    dev = torch.device("cuda", 0)
    src, tgt, T_gt = _c4_pair(a.splats, dev)
    ctxs = {"hem": hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS), "icp": icp_mod.IcpContext(device=0)}
    out = bench.step_c4(ctxs, lru, src, tgt, 0, lambda: torch.cuda.synchronize(dev), a.rank, a.world, comm=comm)
    res = _icp_record(out, T_gt)
    res.update(rank=a.rank, transport=comm.transport, level_sizes=out["level_sizes"], hem_s=out["hem_s"], exchange_s=out["exchange_s"], icp_s=out["icp_s"])
    json.dump(res, open(a.result, "w"))
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


WORKERS = {"c5_single": worker_c5_single, "c5_part": worker_c5_part, "c4_single": worker_c4_single, "c4_part": worker_c4_part}


# ------------------------------------------------------------------------------------------------ parent (never touches the GPU)
def _mock_lib():
    d = os.path.join(ROOT, "tests", "mock_rccl")
    src, lib = os.path.join(d, "mock_rccl.cpp"), os.path.join(d, "libmock_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-std=c++17", src, "-o", lib + ".tmp%d" % os.getpid(), "-lrt"])
        os.replace(lib + ".tmp%d" % os.getpid(), lib)
    return lib


def _run_children(what, a, world, tmp, timeout):
    port = 29800 + (os.getpid() % 150)
    procs, files = [], []
    for r in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if a.transport == "mock" and world > 1:
            env.update(GSR_COMM_TRANSPORT="rccl", GSR_RCCL_LIB=_mock_lib(), GSR_MOCK_RCCL_SLOT_MB=str(a.slot_mb))
        f = os.path.join(tmp, f"{what}_{r}.json")
        files.append(f)
        cmd = [sys.executable, os.path.abspath(__file__), a.mode, "--worker", what, "--rank", str(r), "--world", str(world), "--splats", str(a.splats),
               "--levels", str(a.levels), "--blocks", str(a.world), "--result", f]
        procs.append(subprocess.Popen(cmd, env=env, cwd=ROOT))
    deadline = time.time() + timeout
    rc = []
    for p in procs:
        try:
            rc.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()                  # exactly the children started above
            raise SystemExit(f"{what}: a child did not finish within {timeout:.0f} s")
    if any(rc):
        raise SystemExit(f"{what}: child exit codes {rc}")
    return [json.load(open(f)) for f in files]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["c4", "c5"])
    ap.add_argument("--splats", type=int, default=None)
    ap.add_argument("--world", type=int, default=None)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--transport", choices=["mock", "gloo"], default="mock")
    ap.add_argument("--slot-mb", type=int, default=1024, help="mock RCCL: bytes a rank may publish in one operation (sparse shared memory)")
    ap.add_argument("--timeout", type=float, default=1500.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--worker", default=None)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=None, help="(worker) blocks the cloud is defined by = the world size of the partitioned run")
    ap.add_argument("--result", default=None)
    a = ap.parse_args()
    if a.splats is None:
        a.splats = 40_000_000 if a.mode == "c5" else 5_000_000
    if a.world is None:
        a.world = 8 if a.mode == "c5" else 2
    if a.worker:
        WORKERS[a.worker](a)
        return 0
    t_start = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        single = _run_children(a.mode + "_single", a, 1, tmp, a.timeout)[0]
        ranks = _run_children(a.mode + "_part", a, a.world, tmp, a.timeout)
    rec = {"mode": a.mode, "splats": a.splats, "world": a.world, "levels": a.levels, "transport": ranks[0]["transport"],
           "note": "ranks share ONE MI355X (this pool has no multi-GPU box): byte counts and bit-identity are measured, NO xGMI / collective time exists"}
    ok = True
    if a.mode == "c5":
        M64 = (1 << 64) - 1
        cmp_levels = []
        for k in range(a.levels):
            s = single["levels"][k]
            summed = {f: "%016x" % (sum(int(r["levels"][k]["hash"][f], 16) for r in ranks) & M64) for f in s["hash"]}
            rows = sum(r["levels"][k]["rows"] for r in ranks)
            same = summed == s["hash"] and rows == s["n_out"]
            ok &= same
            cmp_levels.append({"level": k + 1, "n_in_global": s["n_in"], "n_out_global": s["n_out"], "rows_over_ranks": rows, "hash_single": s["hash"],
                               "hash_ranks_summed": summed, "bit_identical": same})
        rec["bit-identical"] = bool(ok)
        rec["compare"] = cmp_levels
        rec["single_context"] = {"peak_device_GB": single["peak_device_GB"],
                                 "levels": [{k: v for k, v in l.items() if k != "hash"} for l in single["levels"]]}
        rec["ranks"] = [{"rank": r["rank"], "peak_device_GB": r["peak_device_GB"], "wall_s_all_levels": r["wall_s_all_levels"],
                         "levels": [{k: v for k, v in l.items() if k != "hash"} for l in r["levels"]]} for r in ranks]
        lv0 = [r["levels"][0] for r in ranks]
        rec["level1_summary"] = {"ghosts_over_owned": sum(l["ghosts"] for l in lv0) / sum(l["owned_in"] for l in lv0),
                                 "halo_MB_received_per_rank_mean": sum(l["halo_bytes_received"] for l in lv0) / len(lv0) / 1e6,
                                 "halo_row_MB_critical_path_per_rank_mean": sum(l["halo_row_bytes_received"] for l in lv0) / len(lv0) / 1e6,
                                 "halo_sh_MB_overlapped_per_rank_mean": sum(l["halo_sh_bytes_received"] for l in lv0) / len(lv0) / 1e6,
                                 "sum_exchange_MB_per_rank_mean": sum(l["sum_exchange_bytes_received"] for l in lv0) / len(lv0) / 1e6,
                                 "bitmap_MB": lv0[0]["bitmap_bytes_allreduced"] / 1e6}
    else:
        import numpy as np
        Ts = np.asarray(single["T"])
        d = [float(np.linalg.norm(np.asarray(r["T"]) - Ts)) for r in ranks]
        same_iters = all(r["iterations"] == single["iterations"] for r in ranks)
        ok = max(d) <= 1e-9 and same_iters
        rec.update({"bit-identical": None, "transform_equal_1e-9": bool(max(d) <= 1e-9), "T_diff_F_vs_single_process": d,
                    "iterations_equal": bool(same_iters), "single_process": single, "ranks": ranks})
    rec["ok"] = bool(ok)
    rec["wall_s"] = time.time() - t_start
    txt = json.dumps(rec, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")
    summary = {k: rec[k] for k in rec if k in ("mode", "splats", "world", "transport", "bit-identical", "transform_equal_1e-9", "iterations_equal", "ok", "wall_s",
                                               "level1_summary", "T_diff_F_vs_single_process")}
    print(json.dumps(summary), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
