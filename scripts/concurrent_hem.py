"""Do the HEM levels of two independent clouds overlap on ONE GPU?  Sequential on one context (what bench.py's step did through round 3)
against two contexts on two streams driven by two host threads (the C ABI releases the GIL; every context owns its workspaces).
usage: python scripts/concurrent_hem.py [n] [reps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingregistration_amd import hem, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 3
A = synth.make_cloud_torch(n, seed=100)
B = synth.make_cloud_torch(n, seed=101)
torch.cuda.synchronize()


def levels(m, c):
    m.set_rng("glibc", 1, 0)
    m.set_level0(c["xyz"], c["color"], c["opacity"], c["cov6"], c["sh"], borrow=True)
    out = []
    for _ in range(L):
        m.run_level()
        out.append(m.size)
    return out


m0 = hem.HemMixture()
for rep in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    a = levels(m0, A); b = levels(m0, B)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"rep{rep} sequential, one context : {dt * 1e3:.2f} ms  {a} {b}", flush=True)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
m1 = hem.HemMixture(stream=s1.cuda_stream)
m2 = hem.HemMixture(stream=s2.cuda_stream)
res = {}
def work(tag, m, c):
    res[tag] = levels(m, c)
for rep in range(reps):
    torch.cuda.synchronize(); t = time.perf_counter()
    th = [threading.Thread(target=work, args=("a", m1, A)), threading.Thread(target=work, args=("b", m2, B))]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"rep{rep} concurrent, two contexts: {dt * 1e3:.2f} ms  {res['a']} {res['b']}", flush=True)
for rep in range(2):       # the same two contexts one after the other (side streams, no overlap): separates the stream effect from the overlap
    torch.cuda.synchronize(); t = time.perf_counter()
    a = levels(m1, A); b = levels(m2, B)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"rep{rep} sequential, two contexts: {dt * 1e3:.2f} ms", flush=True)
