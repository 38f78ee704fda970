"""Randomised sweep of the spatially partitioned HEM (test infrastructure, not collected by pytest): `world` processes over gloo
sharing this box's GPU; random clouds (the shapes of tests/stress_parity.py), two levels; the pieces of all ranks assembled by
global index must equal the single-context levels BIT FOR BIT.  usage: python tests/stress_partition.py [cases] [world] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch.multiprocessing as mp


def clouds(cases, seed):
    from stress_parity import make_case
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(cases):
        desc, c, p = make_case(rng)
        out.append((desc, c))
    return out


def worker(rank, world, port, cases, seed, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussiansplattingregistration_amd import parallel
    from gaussiansplattingregistration_amd.comm import Comm
    cm = Comm.from_torch_group(0)
    res = []
    for desc, c in clouds(cases, seed):
        pieces, st = parallel.hem_partitioned(c, 2, cm, device=0)
        res.append([{k: np.asarray(v) for k, v in p.items()} for p in pieces])
    q.put((rank, res))
    dist.barrier(); dist.destroy_process_group()


def sweep(cases=12, world=3, seed=99, log=print):
    """-> number of clouds whose assembled pieces differ from the one-GPU levels"""
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, 29900 + os.getpid() % 90, cases, seed, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [r[1] for r in sorted((q.get(timeout=1500) for _ in ps), key=lambda x: x[0])]
    [p.join(timeout=120) for p in ps]
    from gaussiansplattingregistration_amd import hem, parallel
    bad = 0
    for i, (desc, c) in enumerate(clouds(cases, seed)):
        want, _ = hem.create_mixture(c, 2)
        ok = True
        for k in range(2):
            got = parallel.assemble_partitioned_level([res[r][i][k] for r in range(world)])
            bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)        # bit patterns: NaN rows must be the same NaNs
            same = got["xyz"].shape == want[k]["xyz"].shape and all(np.array_equal(bits(got[f]), bits(want[k][f])) for f in ("xyz", "color", "cov6", "opacity", "sh"))
            if not same and os.environ.get("STRESS_VERBOSE"):
                print("   level", k + 1, "shapes", got["xyz"].shape, want[k]["xyz"].shape)
                if got["xyz"].shape == want[k]["xyz"].shape:
                    for f in ("xyz", "color", "cov6", "opacity", "sh"):
                        d = np.any(bits(got[f]).reshape(len(got[f]), -1) != bits(want[k][f]).reshape(len(got[f]), -1), axis=1)
                        if d.any():
                            w = np.flatnonzero(d)
                            print("     ", f, "rows differing", len(w), "first", w[:5], "max abs", float(np.abs(np.asarray(got[f], np.float64) - want[k][f]).max()))
            ok = ok and same
        bad += 0 if ok else 1
        log(f"{'ok  ' if ok else 'FAIL'} {i} {desc} {[w['xyz'].shape[0] for w in want]}")
    log(f"{cases - bad} of {cases} partitioned clouds are bit-identical to one GPU (world {world})")
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 99
    sys.exit(1 if sweep(cases, world, seed, log=lambda s: print(s, flush=True)) else 0)


if __name__ == "__main__":
    main()
