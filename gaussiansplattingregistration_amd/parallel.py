"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm, ``gloo`` on CPU test boxes).  The reference has no distributed code at all; this is new design
(SURVEY.md 8e):

* **Two clouds, HEM** -- the clouds are independent: cloud ``A`` on rank 0, cloud ``B`` on rank 1, no
  data-path collective (``assign_clouds``).
* **ICP, any size** -- the target (and its grid) is replicated on every rank, the source points are
  split evenly (``shard_range``); each rank's kernels produce a rank-local accumulator vector
  (32 float64 on the device) and ONE all-reduce(sum) of that device vector per iteration is the only exchange
  (``make_allreduce_device``: RCCL on the tensor where it lies, stream ordered, no host bounce).  It is latency-bound
  (256 bytes), so xGMI link bandwidth is irrelevant; every rank then solves the same 3x3 SVD / 6x6 system ON THE DEVICE
  and stops on the same reduced fitness/RMSE -- the iteration loop stays device resident.
* **One large cloud, HEM** -- spatially partitioned levels (``hem_partitioned``): every rank OWNS one block
  of the cloud (``block_of`` cuts a full cloud; ``synth.make_block_cloud_torch`` draws a rank's block without the rest) and keeps
  only that; per level the library exchanges halo rows with its neighbours (72-byte records on the level's stream, the SH rows on
  a second stream beside the grid phase and the selection) and integer partial sums of the per-child weights -- no float is
  combined across ranks, the levels are bit for bit the one-GPU levels and stay distributed.  (``hem_sharded``, round 2's
  replicated-data work sharding, is kept for comparison only.)
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

__all__ = ["init_distributed", "shard_range", "make_allreduce", "make_allreduce_device", "make_allgather_device", "assign_clouds",
           "registration_icp_sharded", "hem_sharded", "slab_of", "block_dims", "block_of", "hem_partitioned", "assemble_partitioned_level"]


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
    """-> ``fn(buf: np.ndarray[float64])`` that replaces ``buf`` by its sum over the group, in place (host buffers: the
    host-driven ICP loop, kept for callers without device tensors)."""
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


def make_allreduce_device(group=None):
    """-> ``fn(t: CUDA tensor)`` that replaces ``t`` by its sum over the group, in place.  With the ``nccl`` (= RCCL) backend
    the collective runs on the tensor where it lies, ordered on the current stream -- no host round trip.  With ``gloo``
    (CPU test boxes, several ranks sharing one GPU) the tensor is bounced through host memory."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    if dist.get_backend(group) == "nccl":
        def fn(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        def fn(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
    return fn


def make_allgather_device(group=None):
    """-> ``fn(send, recv)``: ``recv`` (world x len(send), CUDA uint8) receives every rank's ``send``."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    if dist.get_backend(group) == "nccl":
        def fn(send, recv):
            dist.all_gather_into_tensor(recv, send, group=group)
    else:
        def fn(send, recv):
            w = dist.get_world_size(group)
            parts = [torch.empty(send.numel(), dtype=send.dtype) for _ in range(w)]
            dist.all_gather(parts, send.cpu(), group=group)
            recv.copy_(torch.cat(parts))
    return fn


def assign_clouds(n_clouds: int, rank: int, world: int):
    """Indices of the independent clouds this rank downsamples (round-robin)."""
    return [i for i in range(n_clouds) if i % world == rank]


def registration_icp_sharded(source, target, max_correspondence_distance, init, estimation_method, criteria,
                             rank: int, world: int, device=None, group=None, ctx=None, comm=None):
    """``registration_icp`` with the source split over the ranks of ``group`` and the target replicated.

    ``comm`` (a ``comm.Comm``): the collective is the library's own -- ncclAllReduce enqueued on the ICP context's stream,
    no Python per iteration; without it a ``torch.distributed`` trampoline is installed (kept for callers without a ``Comm``).

    ``source`` / ``target`` are the FULL clouds on every rank (``PointCloud`` records); each rank keeps its
    ``shard_range`` of the source points (and of their covariances / colours).  The only exchange is one all-reduce of
    32 float64 per iteration, on the device; every rank solves from the reduced vector.  A rank whose shard is empty
    (fewer points than ranks) contributes zeros.  Returns the same ``RegistrationResult`` on every rank."""
    from .models.point_cloud import PointCloud
    from .utils.local_registration_util import registration_icp

    n = len(source)
    lo, hi = shard_range(n, rank, world)
    cut = lambda a: None if a is None else a[lo:hi]
    local = PointCloud(xyz32=source.xyz32[lo:hi], colors=cut(source.colors), cov6=cut(source.cov6))
    if comm is not None:          # the library's own communicator (RCCL enqueued on the context's stream)
        return registration_icp(local, target, max_correspondence_distance, init, estimation_method, criteria,
                                device=device, comm=comm, n_source_global=n, ctx=ctx)
    ar = make_allreduce_device(group)
    return registration_icp(local, target, max_correspondence_distance, init, estimation_method, criteria,
                            device=device, allreduce_device=ar, n_source_global=n, ctx=ctx)


def hem_sharded(cloud: dict, cluster_level: int, rank: int, world: int, device=None, group=None, allreduce=None, allgather=None,
                as_torch=False, **hem_params):
    """``MixtureCreator.CreateMixture`` of ONE large cloud with the work of every level split over ``world``
    GPUs (BASELINE config 5).  Every rank passes the same full ``cloud`` (replicated data); rank r evaluates
    the r-th spatial slab of parents; two exchanges per level: an all-reduce of the per-child sums (float32[n]) and one
    all-gather of the merged components (packed rows).  Every rank returns the identical list of levels."""
    from . import hem as _hem
    if allreduce is None and world > 1:
        allreduce = make_allreduce_device(group)
    if allgather is None and world > 1:
        allgather = make_allgather_device(group)
    dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    with _hem.HemMixture(device=dev, **hem_params) as m:
        m.set_shard(rank, world, allreduce, allgather)
        m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"])
        levels, stats = [], []
        for _ in range(int(cluster_level)):
            m.run_level()
            stats.append(m.stats())
            levels.append(m.get_level(as_torch=as_torch))
        return levels, stats


def slab_of(xyz, rank: int, world: int):
    """Global indices (ascending) of the components of slab ``rank`` of ``world``: the cloud cut along its longest axis into
    ``world`` slabs of equal counts (a stable argsort of that coordinate, so every rank computes the same cut)."""
    if torch is not None and isinstance(xyz, torch.Tensor):
        ext = xyz.max(0).values - xyz.min(0).values
        axis = int(torch.argmax(ext))
        order = torch.argsort(xyz[:, axis], stable=True)
        lo, hi = shard_range(xyz.shape[0], rank, world)
        return torch.sort(order[lo:hi]).values
    x = np.asarray(xyz)
    axis = int(np.argmax(x.max(0) - x.min(0)))
    order = np.argsort(x[:, axis], kind="stable")
    lo, hi = shard_range(x.shape[0], rank, world)
    return np.sort(order[lo:hi])


def block_dims(world: int):
    """(px, py, pz) with px * py * pz == world and the factors as equal as possible (8 -> 2 x 2 x 2, 4 -> 2 x 2 x 1, 6 -> 3 x 2 x 1)."""
    best = (world, 1, 1)
    for a in range(1, world + 1):
        if world % a:
            continue
        for b in range(1, world // a + 1):
            if (world // a) % b:
                continue
            d = tuple(sorted((a, b, world // a // b), reverse=True))
            if max(d) - min(d) < max(best) - min(best):
                best = d
    return best


def block_of(xyz, rank: int, world: int):
    """Global indices (ascending) of the components of block ``rank`` of ``world``: the cloud cut into px x py x pz blocks of equal
    counts (``block_dims``) -- px slabs along the longest axis, every slab into py columns along the second, every column into pz
    blocks along the third; stable sorts, so every rank computes the same cut.  A block has less surface than a slab: with the
    bench cloud's search radii a rank of 8 receives halo rows worth 40 % of its own components instead of 99 % (40 M splats,
    scripts/halo_estimate.py).  world = 2, 3, 5, 7: the same as ``slab_of``."""
    tt = torch is not None and isinstance(xyz, torch.Tensor)
    x = xyz if tt else np.asarray(xyz)
    ext = (x.max(0).values - x.min(0).values) if tt else (x.max(0) - x.min(0))
    axes = [int(a) for a in (torch.argsort(ext, descending=True) if tt else np.argsort(-ext))]
    dims = block_dims(world)
    coord = [rank // (dims[1] * dims[2]), (rank // dims[2]) % dims[1], rank % dims[2]]
    idx = torch.arange(x.shape[0], device=x.device) if tt else np.arange(x.shape[0])
    for level in range(3):
        if dims[level] == 1:
            continue
        v = x[idx, axes[level]]
        order = torch.argsort(v, stable=True) if tt else np.argsort(v, kind="stable")
        lo, hi = shard_range(idx.shape[0], coord[level], dims[level])
        idx = idx[order[lo:hi]]
    return torch.sort(idx).values if tt else np.sort(idx)


def hem_partitioned(cloud: dict, cluster_level: int, comm, device=None, as_torch=False, owned=None, mixture=None, n_global=None,
                    **hem_params):
    """``MixtureCreator.CreateMixture`` of ONE large cloud SPATIALLY partitioned over the ranks of ``comm`` (BASELINE config 5):
    this rank keeps the components of its block (``owned`` = their global indices; default ``block_of``) -- here cut out of the full
    ``cloud`` every rank passes; a caller that holds only its slab passes that and ``owned`` -- and every level runs on owned +
    halo components, bit for bit the single-GPU level (include/gsr_hip.h, gsr_hem_set_level0_part).  Returns (pieces, stats): per
    level a dict of this rank's rows and ``gid``, their positions in the level's global order; ``assemble_partitioned_level``
    puts the pieces of all ranks together.

    ``n_global`` given: ``cloud`` IS this rank's block already (``owned`` = its ascending global indices, required) and nothing is
    cut out -- the caller never held the whole cloud (``synth.make_block_cloud_torch``, bench.py --mode c5).

    A rank whose call raises leaves its peers inside the next collective (include/gsr_hip.h, "errors are local"): the caller must
    tear the process group down -- under ``torch.distributed.run`` the uncaught exception does."""
    from . import hem as _hem
    dev = device if device is not None else (torch.cuda.current_device() if torch.cuda.is_available() else 0)
    if n_global is not None:
        if owned is None:
            raise RuntimeError("hem_partitioned: a local block (n_global given) needs its global indices in `owned`")
        idx = owned
        take = lambda a: a
    else:
        n_global = int(cloud["xyz"].shape[0])
        idx = owned if owned is not None else block_of(cloud["xyz"], comm.rank, comm.world)
        take = lambda a: a[idx]
    n_global = int(n_global)
    import contextlib
    # a caller-provided mixture context keeps its workspaces from call to call (no allocation in steady state)
    with (contextlib.nullcontext(mixture) if mixture is not None else _hem.HemMixture(device=dev, **hem_params)) as m:
        if mixture is not None:
            m.set_rng(hem_params.get("rng_mode", "glibc"), hem_params.get("rng_seed", 1), hem_params.get("rng_skip", 0))
        m.set_comm(comm)
        m.set_level0_part(take(cloud["xyz"]), take(cloud["color"]), take(cloud["opacity"]), take(cloud["cov6"]), take(cloud["sh"]), idx, n_global)
        pieces, stats = [], []
        for _ in range(int(cluster_level)):
            m.run_level()
            st = m.stats()
            st.update(m.part_stats())
            stats.append(st)
            piece = m.get_level(as_torch=as_torch)
            piece["gid"] = m.gids()
            pieces.append(piece)
        m.set_comm(None)
        return pieces, stats


def assemble_partitioned_level(pieces_of_all_ranks):
    """The global level out of every rank's piece (host arrays): row gid[k] of the result = row k of the piece."""
    n = sum(len(p["gid"]) for p in pieces_of_all_ranks)
    out = {}
    for f in ("xyz", "color", "cov6", "opacity", "sh"):
        first = np.asarray(pieces_of_all_ranks[0][f])
        full = np.empty((n,) + first.shape[1:], first.dtype)
        for p in pieces_of_all_ranks:
            full[np.asarray(p["gid"], np.int64)] = np.asarray(p[f])
        out[f] = full
    return out
