"""GSR_ICP_ADAPT (a finer target grid when the points are clumped) against the box-volume rule: the ICP schedule on the levels of a 5 M pair,
clustered and isotropic.  usage: python scripts/icp_adapt_ab.py [n]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for shape in ("clustered", "iso"):
    for adapt in ("0", "1"):
        env = dict(os.environ, GSR_ICP_ADAPT=adapt)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", shape, "--no-cpu-baseline", "--no-aniso", "--steps", "3"], env=env, capture_output=True, text=True)
        rows = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not rows:
            print(shape, adapt, "FAILED", r.stderr[-400:]); continue
        d = json.loads(rows[-1])
        print(f"{shape} adapt={adapt}: step {d['ms_per_step']:.2f} ms, icp {d['icp_s_per_step'] * 1e3:.2f} ms, T_err {d['icp_result']['T_err_vs_ground_truth_F']:.1e}, per level "
              + ", ".join(f"{l['ns']}: {l['iterations']} x {l['ms_per_iteration']:.3f} (+{l['ms_target_index_build']:.2f})" for l in d["icp_per_level"]), flush=True)
