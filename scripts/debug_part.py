import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_distributed_gpu as T
from gaussiansplattingregistration_amd import hem, parallel, synth
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2

def main():
    res = T._run("part", world)
    for tag, c in (("iso", synth.make_cloud(60000, seed=61, sh_degree=1)), ("aniso", synth.make_cloud(40000, seed=62, sh_degree=0, shape="aniso"))):
        if tag == "iso":
            c["xyz"][7] = [9.0, 0.5, -0.5]; c["cov6"][7] = [4.0, 0, 0, 3.0, 0, 2.0]
            c["cov6"][11] = [1.0, 0, 0, 1.0, 0, -1.0]
        want, wst = hem.create_mixture(c, 3)
        for k in range(3):
            got = parallel.assemble_partitioned_level([res[r][tag][k] for r in range(world)])
            st = [res[r][tag + "_stats"][k] for r in range(world)]
            print(tag, k, "n", got["xyz"].shape[0], want[k]["xyz"].shape[0], "parents", [s["parents"] for s in st], wst[k]["parents"], "pairs", sum(s["pairs"] for s in st), wst[k]["pairs"],
                  "orphans", sum(s["orphans"] for s in st), wst[k]["orphans"], "ghosts", [s["ghosts"] for s in st])
            if got["xyz"].shape != want[k]["xyz"].shape:
                continue
            for f in ("xyz", "color", "cov6", "opacity", "sh"):
                a, b = got[f].reshape(got[f].shape[0], -1), want[k][f].reshape(want[k][f].shape[0], -1)
                neq = (a != b).any(1)
                print("   ", f, "rows differing", int(neq.sum()), "max abs diff", float(np.abs(a.astype(np.float64) - b).max()), "first bad rows", np.flatnonzero(neq)[:8])
            # are the differing xyz rows of the parents' part or the orphans' part?
            neq = (got["xyz"] != want[k]["xyz"]).any(1)
            P = wst[k]["parents"]
            print("    differing among parents", int(neq[:P].sum()), "among orphans", int(neq[P:].sum()))


if __name__ == "__main__":
    main()
