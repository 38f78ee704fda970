import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.multiprocessing as mp


def cloud(variant):
    from gaussiansplattingregistration_amd import synth
    c = synth.make_cloud(60000, seed=61, sh_degree=1)
    if variant >= 1:
        c["xyz"][7] = [9.0, 0.5, -0.5]; c["cov6"][7] = [4.0, 0, 0, 3.0, 0, 2.0]
    if variant >= 2:
        c["cov6"][11] = [1.0, 0, 0, 1.0, 0, -1.0]
    return c


def worker(rank, world, port, q, variant):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from gaussiansplattingregistration_amd import parallel, hem as _hem
    from gaussiansplattingregistration_amd.comm import Comm
    parallel.init_distributed("gloo")
    torch.cuda.set_device(0)
    cm = Comm.from_torch_group(0)
    c = cloud(variant)
    idx = parallel.slab_of(c["xyz"], rank, world)
    with _hem.HemMixture(device=0) as m:
        m.set_comm(cm)
        m.set_level0_part(c["xyz"][idx], c["color"][idx], c["opacity"][idx], c["cov6"][idx], c["sh"][idx], idx, 60000)
        m.run_level()
        st = m.stats(); st.update(m.part_stats())
        p = m.get_level(with_state=True); p["gid"] = m.gids()
    q.put((rank, p, st))
    dist.barrier(); dist.destroy_process_group()


def main():
    from gaussiansplattingregistration_amd import hem
    for variant in (0, 1, 2):
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, 2, 29700 + variant, q, variant)) for r in range(2)]
        [p.start() for p in procs]
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
        [p.join() for p in procs]
        c = cloud(variant)
        with hem.HemMixture(device=0) as m:
            m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"])
            m.run_level(); wst = m.stats(); want = m.get_level(with_state=True)
        n = want["xyz"].shape[0]
        print("variant", variant, "cells", wst["cells"], [r[2]["cells"] for r in res], "irregular", wst["irregular"], [r[2]["irregular"] for r in res],
              "ghosts", [r[2]["ghosts"] for r in res], "cand", wst["candidates"], sum(r[2]["candidates"] for r in res))
        for f in ("xyz", "weight", "opacity"):
            full = np.empty_like(want[f])
            for r in res:
                full[r[1]["gid"].astype(np.int64)] = r[1][f]
            a = full.reshape(n, -1); b = want[f].reshape(n, -1)
            neq = (a != b).any(1)
            print("   ", f, "rows differing", int(neq.sum()), "of", n, "max rel", float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-30)))


if __name__ == "__main__":
    main()
