"""debug: where do the partitioned level-1 rows differ from the single-context ones? usage: debug_c5.py N WORLD"""
import os, sys, subprocess, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SEED = 100

def worker(kind, n, world, rank, outdir):
    import torch
    import bench
    from gaussiansplattingregistration_amd import hem, synth, parallel
    dev = torch.device("cuda", 0)
    if kind == "single":
        parts = [synth.make_block_cloud_torch(n, r, world, seed=SEED, device=dev)[0] for r in range(world)]
        cloud = {k: torch.cat([p[k] for p in parts]).contiguous() for k in ("xyz", "color", "opacity", "cov6", "sh")}
        with hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS) as m:
            m.set_level0(cloud["xyz"], cloud["color"], cloud["opacity"], cloud["cov6"], cloud["sh"], borrow=True)
            lv0 = m.get_level(as_torch=True, with_state=True)
            np.save(f"{outdir}/single_par0.npy", lv0["is_parent"].cpu().numpy())
            m.run_level()
            st = m.stats()
            d = m.get_level(as_torch=True, with_state=True)
            np.save(f"{outdir}/single_xyz.npy", d["xyz"].cpu().numpy()); np.save(f"{outdir}/single_w.npy", d["weight"].cpu().numpy())
            np.save(f"{outdir}/in_xyz.npy", cloud["xyz"].cpu().numpy())
            json.dump(st, open(f"{outdir}/single_st.json", "w"))
        return
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussiansplattingregistration_amd.comm import Comm
    comm = Comm.from_torch_group(0)
    blk, gid0 = synth.make_block_cloud_torch(n, rank, world, seed=SEED, device=dev)
    m = hem.HemMixture(device=0, rng_mode="glibc", **bench.HEM_PARAMS)
    m.set_comm(comm)
    m.set_level0_part(blk["xyz"], blk["color"], blk["opacity"], blk["cov6"], blk["sh"], gid0, n)
    m.run_level()
    st = m.stats(); st.update(m.part_stats())
    d = m.get_level(as_torch=True, with_state=True)
    np.save(f"{outdir}/part_xyz_{rank}.npy", d["xyz"].cpu().numpy()); np.save(f"{outdir}/part_w_{rank}.npy", d["weight"].cpu().numpy())
    np.save(f"{outdir}/part_gid_{rank}.npy", m.gids())
    json.dump(st, open(f"{outdir}/part_st_{rank}.json", "w"))
    dist.barrier(); m.set_comm(None); m.close(); comm.close(); dist.destroy_process_group()

def main():
    if len(sys.argv) > 3 and sys.argv[3] in ("single", "part"):
        worker(sys.argv[3], int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4]), sys.argv[5]); return
    n, world = int(sys.argv[1]), int(sys.argv[2])
    outdir = "/tmp/dbgc5"; os.makedirs(outdir, exist_ok=True)
    subprocess.check_call([sys.executable, __file__, str(n), str(world), "single", "0", outdir], cwd=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29877", WORLD_SIZE=str(world))
    if os.environ.get("DBG_MOCK", "1") == "1":
        env.update(GSR_COMM_TRANSPORT="rccl", GSR_RCCL_LIB=os.path.join(ROOT, "tests/mock_rccl/libmock_rccl.so"), GSR_MOCK_RCCL_SLOT_MB="1024")
    ps = [subprocess.Popen([sys.executable, __file__, str(n), str(world), "part", str(r), outdir], cwd=ROOT, env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(world)]
    assert all(p.wait() == 0 for p in ps)
    sst = json.load(open(f"{outdir}/single_st.json"))
    P = sst["parents"]
    sx, sw = np.load(f"{outdir}/single_xyz.npy"), np.load(f"{outdir}/single_w.npy")
    gx = np.full((max(sx.shape[0], 1) + 100000, 3), np.nan, np.float32); gw = np.full(gx.shape[0], np.nan, np.float32)
    tot = 0
    for r in range(world):
        g = np.load(f"{outdir}/part_gid_{r}.npy").astype(np.int64); gx[g] = np.load(f"{outdir}/part_xyz_{r}.npy"); gw[g] = np.load(f"{outdir}/part_w_{r}.npy"); tot += len(g)
        print("rank", r, {k: v for k, v in json.load(open(f"{outdir}/part_st_{r}.json")).items() if k in ("parents", "pairs", "orphans", "ghosts", "irregular", "heavy_parents", "n_in", "n_out")})
    print("single", {k: sst[k] for k in ("parents", "pairs", "orphans", "irregular", "heavy_parents", "n_out")}, "part rows", tot)
    # parents: the first P rows of either (input order of the parents)
    par0 = np.load(f"{outdir}/single_par0.npy").astype(bool)
    inx = np.load(f"{outdir}/in_xyz.npy")
    pidx = np.nonzero(par0)[0]
    assert len(pidx) == P
    dif = np.nonzero((sx[:P].view(np.uint32) != gx[:P].view(np.uint32)).any(1) | (sw[:P].view(np.uint32) != gw[:P].view(np.uint32)))[0]
    print("parents whose merged row differs:", len(dif), "of", P)
    if len(dif):
        from gaussiansplattingregistration_amd import synth
        h = synth.half_extent(n)
        pos = inx[pidx[dif]]
        print("weight single - part: ", np.percentile(sw[dif] - gw[dif], [0, 25, 50, 75, 100]))
        # distance of the differing parents to the nearest internal block face (2x2x2 blocks: the planes x=0,y=0,z=0) and to the outer box
        print("|coord| to internal faces (min over axes) percentiles:", np.percentile(np.abs(pos).min(1), [0, 25, 50, 75, 100]))
        print("distance to the outer box percentiles:", np.percentile((h - np.abs(pos)).min(1), [0, 25, 50, 75, 100]), "h", h)
        allp = inx[pidx]
        print("for ALL parents: |coord| min-axis median", np.median(np.abs(allp).min(1)), "outer median", np.median((h - np.abs(allp)).min(1)))
        print("first differing parents (input index, pos, w_single, w_part):")
        for k in dif[:10]: print(pidx[k], inx[pidx[k]], sw[k], gw[k])
main()
