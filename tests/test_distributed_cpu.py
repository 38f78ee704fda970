"""N>1 path on CPU: two gloo ranks split the source, all-reduce the accumulator vector through the
product's parallel.make_allreduce, solve on every rank with the product's host solve, and land on the
single-process answer.  (Rank-local accumulators come from a NumPy stand-in of the kernel here; the
kernel itself is checked against the oracle in the -m gpu tests.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _acc_numpy(src, tgt, T, max_corr, centre):
    from scipy.spatial import cKDTree
    p = src @ T[:3, :3].T + T[:3, 3]
    d, j = cKDTree(tgt).query(p, k=1)
    m = d * d < max_corr * max_corr
    a, b = p[m] - centre, tgt[j[m]] - centre
    acc = np.zeros(32)
    acc[0], acc[1] = m.sum(), (d[m] ** 2).sum()
    acc[2:5], acc[5:8] = a.sum(0), b.sum(0)
    acc[8:17] = (a[:, :, None] * b[:, None, :]).sum(0).reshape(-1)
    return acc


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gaussiansplattingregistration_amd import _lib, parallel, synth
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    src, tgt, T_gt = synth.make_pair(3000, seed=3, sh_degree=0)
    s, t = src["xyz"].astype(np.float64), tgt["xyz"].astype(np.float64)
    lo, hi = parallel.shard_range(len(s), rank, world)
    centre = 0.5 * (t.min(0) + t.max(0))
    allreduce = parallel.make_allreduce()
    L = _lib.load()
    T = np.eye(4)
    for _ in range(6):
        acc = _acc_numpy(s[lo:hi], t, T, 0.25, centre)
        allreduce(acc)                                      # the ONLY exchange per iteration
        upd = np.zeros((4, 4))
        assert L.gsr_icp_solve(acc.ctypes.data, 0, centre.ctypes.data, upd.ctypes.data) == 0
        T = upd @ T
    q.put((rank, T, acc[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_source_split_equals_single_process(hip_lib, oracle):
    from gaussiansplattingregistration_amd import parallel, synth
    assert parallel.shard_range(10, 0, 3) == (0, 4) and parallel.shard_range(10, 2, 3) == (7, 10)
    assert parallel.assign_clouds(2, 1, 2) == [1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    assert np.array_equal(res[0][1], res[1][1])             # every rank holds the identical transform
    assert res[0][2] == res[1][2] == 3000                   # reduced correspondence count is global
    src, tgt, T_gt = synth.make_pair(3000, seed=3, sh_degree=0)
    want = oracle.icp(src["xyz"], tgt["xyz"], None, np.eye(4), kind=0, max_corr=0.25, max_iter=6, rel_fitness=0, rel_rmse=0)
    assert np.linalg.norm(res[0][1] - want["transformation"]) < 1e-9


def test_block_partition_covers_the_cloud_once():
    """parallel.block_of / slab_of: every component belongs to exactly one rank, counts differ by at most one per cut, numpy and
    torch inputs give the same cut, and the blocks are boxes (8 ranks: 2 x 2 x 2; the halo of a block is 40 % of its own components
    at 40 M splats where a slab's is 99 %, scripts/halo_estimate.py)."""
    import torch
    from gaussiansplattingregistration_amd import parallel
    assert [parallel.block_dims(w) for w in (1, 2, 3, 4, 6, 8, 16)] == [(1, 1, 1), (2, 1, 1), (3, 1, 1), (2, 2, 1), (3, 2, 1), (2, 2, 2), (4, 2, 2)]
    rng = np.random.default_rng(3)
    x = (rng.uniform(-1, 1, (5003, 3)) * [3.0, 2.0, 1.0]).astype(np.float32)
    for fn in (parallel.block_of, parallel.slab_of):
        for w in (1, 2, 3, 4, 8):
            parts = [fn(x, r, w) for r in range(w)]
            assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(5003))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 3
            assert all(np.array_equal(fn(torch.from_numpy(x), r, w).numpy(), parts[r]) for r in range(w))
    parts = [parallel.block_of(x, r, 8) for r in range(8)]
    boxes = [(x[p].min(0), x[p].max(0)) for p in parts]
    for a in range(8):
        for b in range(a + 1, 8):      # two blocks are separated along at least one axis
            assert any(boxes[a][1][k] <= boxes[b][0][k] or boxes[b][1][k] <= boxes[a][0][k] for k in range(3)), (a, b)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8])
def test_block_clouds_tile_the_scene(world):
    """bench.py --mode c5 never materialises the large cloud: it is DEFINED block by block (synth.make_block_cloud_torch).  The blocks'
    boxes tile the cube without gaps or overlaps, their global indices are the contiguous shard ranges (ascending, disjoint, n in all),
    every block's splats lie inside its own box, and a rank's block does not depend on who else draws theirs (CPU tensors here)."""
    import torch
    from gaussiansplattingregistration_amd import parallel, synth
    n = 12_345
    h = synth.half_extent(n)
    vol, gids = 0.0, []
    for r in range(world):
        lo, hi = synth.block_box(r, world, h)
        assert (hi > lo).all() and (lo >= -h - 1e-9).all() and (hi <= h + 1e-9).all()
        vol += float(np.prod(hi - lo))
        c, gid = synth.make_block_cloud_torch(n, r, world, seed=3, device="cpu", sh_degree=1)
        a, b = parallel.shard_range(n, r, world)
        assert gid.dtype == torch.int32 and gid.tolist() == list(range(a, b)) and c["xyz"].shape == (b - a, 3) and c["sh"].shape == (b - a, 9)
        x = c["xyz"].double().numpy()
        assert (x >= lo - 1e-5).all() and (x <= hi + 1e-5).all()
        c2, _ = synth.make_block_cloud_torch(n, r, world, seed=3, device="cpu", sh_degree=1)
        assert torch.equal(c["xyz"], c2["xyz"]) and torch.equal(c["cov6"], c2["cov6"])
        gids.append(gid)
    assert abs(vol - (2 * h) ** 3) < 1e-6 * (2 * h) ** 3
    assert torch.equal(torch.cat(gids), torch.arange(n, dtype=torch.int32))
    # boxes of different ranks do not overlap (interiors)
    for r in range(world):
        for q in range(r + 1, world):
            lo1, hi1 = synth.block_box(r, world, h)
            lo2, hi2 = synth.block_box(q, world, h)
            assert ((np.minimum(hi1, hi2) - np.maximum(lo1, lo2)) <= 1e-9).any()
