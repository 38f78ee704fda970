"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm, ``gloo`` on CPU test boxes).  The reference has no distributed code at all; this is new design
(SURVEY.md 8e):

* **Two clouds, HEM** -- the clouds are independent: cloud ``A`` on rank 0, cloud ``B`` on rank 1, no
  data-path collective (``assign_clouds``).
* **ICP, any size** -- the target (and its grid) is replicated on every rank, the source points are
  split evenly (``shard_range``); each rank's fused kernel produces a rank-local accumulator vector
  (32 float64) and ONE all-reduce(sum) of that vector per iteration is the only exchange
  (``make_allreduce``).  It is latency-bound (256 bytes), so xGMI link bandwidth is irrelevant;
  every rank then solves the same 3x3 SVD / 6x6 system and stops on the same reduced fitness/RMSE.
* **One large cloud, HEM** -- work-sharded levels (``hem_sharded``): the level data is replicated, the
  cell-sorted parents are split into ``world`` contiguous runs (spatial slabs), and a level makes two
  all-reduces: the per-child sums of wL (float32[n]) and the merged components (each row written by one
  rank, zeros elsewhere, so the sum is exact).  Flags, orphans and the validity erase are computed
  identically on every rank.
"""
from __future__ import annotations

import os

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

__all__ = ["init_distributed", "shard_range", "make_allreduce", "assign_clouds", "registration_icp_sharded", "hem_sharded"]


def init_distributed(backend: str | None = None):
    """Initialise the default process group from the launcher's environment (RANK, WORLD_SIZE, MASTER_*).
    Returns (rank, world_size, local_rank).  Single-process runs return (0, 1, 0) without a group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = os.environ.get("GSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(n: int, rank: int, world: int):
    """Contiguous, near-even split of ``n`` items: the first ``n % world`` ranks get one extra."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_allreduce(group=None, device=None):
    """-> ``fn(buf: np.ndarray[float64])`` that replaces ``buf`` by its sum over the group, in place."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    use_cuda = dist.get_backend(group) == "nccl"

    def fn(buf: np.ndarray):
        t = torch.from_numpy(buf.copy())
        if use_cuda:
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        buf[:] = t.cpu().numpy()

    return fn


def assign_clouds(n_clouds: int, rank: int, world: int):
    """Indices of the independent clouds this rank downsamples (round-robin)."""
    return [i for i in range(n_clouds) if i % world == rank]


def registration_icp_sharded(source, target, max_correspondence_distance, init, estimation_method, criteria,
                             rank: int, world: int, device=None, group=None):
    """``registration_icp`` with the source split over the ranks of ``group`` and the target replicated.

    ``source`` / ``target`` are the FULL clouds on every rank (``PointCloud`` records); each rank keeps
    its ``shard_range`` of the source.  Returns the same ``RegistrationResult`` on every rank."""
    from .models.point_cloud import PointCloud
    from .utils.local_registration_util import registration_icp

    n = len(source)
    lo, hi = shard_range(n, rank, world)
    local = PointCloud(xyz32=source.xyz32[lo:hi])
    ar = make_allreduce(group, None if device is None else torch.device("cuda", device))
    return registration_icp(local, target, max_correspondence_distance, init, estimation_method, criteria,
                            device=device, allreduce=ar, n_source_global=n)


def hem_sharded(cloud: dict, cluster_level: int, rank: int, world: int, device=None, group=None, allreduce=None,
                as_torch=False, **hem_params):
    """``MixtureCreator.CreateMixture`` of ONE large cloud with the work of every level split over ``world``
    GPUs (BASELINE config 5).  Every rank passes the same full ``cloud`` (replicated data); rank r evaluates
    the r-th spatial slab of parents; two RCCL all-reduces per level (per-child sums, merged components).
    Every rank returns the identical list of levels.  ``allreduce(tensor)`` defaults to
    ``torch.distributed.all_reduce`` on ``group``."""
    from . import hem as _hem
    if allreduce is None and world > 1:
        def allreduce(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    with _hem.HemMixture(device=dev, **hem_params) as m:
        m.set_shard(rank, world, allreduce)
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        levels, stats = [], []
        for _ in range(int(cluster_level)):
            m.run_level()
            stats.append(m.stats())
            levels.append(m.get_level(as_torch=as_torch))
        return levels, stats
