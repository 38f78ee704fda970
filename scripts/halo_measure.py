"""Ghost components of a spatially partitioned HEM level, measured: `world` processes over gloo sharing this box's GPU (the callback
transport), a cloud of n splats at the bench density cut into blocks (parallel.block_of); prints, per rank and level, the ghosts
received, the rows sent and the bytes of the halo exchange -- with the halo marked by the pre-reject ellipsoid's box (default) and by
the search sphere's box (GSR_HEM_ELL=0).  usage: python scripts/halo_measure.py [n] [world]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp


def worker(rank, world, port, n, q):
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussiansplattingregistration_amd import parallel, synth
    from gaussiansplattingregistration_amd.comm import Comm
    cm = Comm.from_torch_group(0)
    c = synth.make_cloud(n, seed=5, sh_degree=3)
    pieces, st = parallel.hem_partitioned(c, 2, cm, device=0)
    q.put((rank, [(s["n_in"], s["ghosts"], s["rows_sent"], s["halo_bytes_received"], s["sum_exchange_bytes_received"]) for s in st]))
    dist.barrier(); dist.destroy_process_group()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for ell in ("1", "0"):
        os.environ["GSR_HEM_ELL"] = ell
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, 29700 + (os.getpid() % 200) + int(ell), n, q)) for r in range(world)]
        [p.start() for p in ps]
        res = sorted(q.get(timeout=900) for _ in ps)
        [p.join(timeout=120) for p in ps]
        for lvl in range(2):
            own = sum(r[1][lvl][0] for r in res); gh = sum(r[1][lvl][1] for r in res); by = sum(r[1][lvl][3] for r in res)
            print(f"n={n} world={world} halo by {'ellipsoid box' if ell == '1' else 'sphere box'} level {lvl + 1}: own {own}, ghosts {gh} = {gh / own:.1%} of own, "
                  f"halo bytes received per rank {by / world / 1e6:.1f} MB", flush=True)


if __name__ == "__main__":
    main()
