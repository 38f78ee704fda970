#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts/gpu_round.sh into the small tracked files under profiles/.
The kernel-stats table comes from the bench.py run; the PMC passes profile scripts/prof_hem.py 5000000 1 (level 1 of a 5 M
cloud alone), and the summary records the hash of the kernel sources it was taken with (bench.py only quotes it when the
sources still match).

usage: python scripts/summarize_profiles.py gpurun_out/r01 profiles/archive/r01
Writes <dst>_kernel_stats.csv (the --stats summary, kernel names shortened), <dst>_pmc.json (per-kernel
counter sums / launch, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md applied where
stated) and copies the bench JSON lines."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def newest(pattern):
    """gpurun MERGES every call's output into the local directory, so a pass's directory can hold the files of several runs: only the
    newest run's counter file counts (one file per profiled process)."""
    fs = glob.glob(pattern)
    return [max(fs, key=os.path.getmtime)] if fs else []


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"(radix_sort_\w+|scan_\w+|lookback_scan\w*|transform_\w+|init_\w+)", name)
    if "rocprim" in name and m:
        return "rocprim::" + m.group(1)
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    return name[:90]


f = newest(os.path.join(src, "prof_stats", "*", "*kernel_stats.csv"))
if f:
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.OrderedDict()
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]); a[2] = min(a[2], float(r["MinNs"])); a[3] = max(a[3], float(r["MaxNs"]))
    tot = sum(a[1] for a in agg.values())
    with open(dst + "_kernel_stats.csv", "w") as o:
        o.write("kernel,calls,total_ms,avg_ms,pct,min_ms,max_ms\n")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{k},{a[0]},{a[1]/1e6:.3f},{a[1]/a[0]/1e6:.4f},{100*a[1]/tot:.2f},{a[2]/1e6:.4f},{a[3]/1e6:.4f}\n")



def derived(e, v, n):
    """Derived figures of one kernel's counters (per launch): lane utilisation, cycles per VALU instruction, wait fractions, VALU busy."""
    g = lambda k: v.get(k, 0.0) / n
    if g("SQ_ACTIVE_INST_VALU") > 0:
        e["lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        if g("SQ_INSTS_VALU") > 0:
            e["cycles_per_valu_inst"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
        if g("SQ_BUSY_CYCLES") > 0:
            e["valu_busy_frac"] = (4.0 * g("SQ_ACTIVE_INST_VALU") / 1024.0) / (g("SQ_BUSY_CYCLES") / 32.0)
    if g("SQ_WAVE_CYCLES") > 0:
        e["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        e["wait_inst_any_frac_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAVES") > 0 and g("SQ_INSTS_VALU") > 0:
        e["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")

pmc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int)
for d in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_sq2", "pmc_tc"):
    for f in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("gsr::"):
                continue
            pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (d, k, r["Dispatch_Id"])
            if d == "pmc_fetch" and key not in seen:
                seen.add(key); launches[k] += 1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_build_id: the kernel sources these counters belong to)
out = {"kernel_build": bench.kernel_build_id(),
       "command": "scripts/prof_hem.py 5000000 1 2  (ONE level of a 5 M-splat cloud, SH degree 3: the level-1 launch alone; per-launch averages)"}
for k, v in pmc.items():
    n = max(1, launches.get(k, 1))
    e = {"launches_in_pass": n}
    for c, val in v.items():
        e[c + "_per_launch"] = val / n
    if "FETCH_SIZE" in v:
        # FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
        e["hbm_read_bytes_per_launch_x2_corrected"] = v["FETCH_SIZE"] * 1024 * 2 / n
        e["hbm_read_bytes_per_launch_raw"] = v["FETCH_SIZE"] * 1024 / n
    if "WRITE_SIZE" in v:
        e["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024 / n
    if "TCC_HIT_sum" in v:
        e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    derived(e, v, n)
    out[k] = e
# the whole level: every kernel's launches of the pass added up, per repetition of the level (the pass runs scripts/prof_hem.py ... 2: the
# level twice; a level's prologue runs with its input and the next level's behind its last kernel -- both are in the sum)
REPS = 2
tot_r = sum(e.get("hbm_read_bytes_per_launch_raw", 0.0) * e["launches_in_pass"] for k, e in out.items() if isinstance(e, dict)) / REPS
tot_w = sum(e.get("hbm_write_bytes_per_launch", 0.0) * e["launches_in_pass"] for k, e in out.items() if isinstance(e, dict)) / REPS
out["level_total"] = {"repetitions_in_pass": REPS, "hbm_read_bytes_raw": tot_r, "hbm_read_bytes_x2_corrected": 2.0 * tot_r, "hbm_write_bytes": tot_w,
                      "traffic_raw": tot_r + tot_w, "traffic_x2": 2.0 * tot_r + tot_w,
                      "note": "sum over every gsr:: kernel of one 5 M-splat level (FETCH_SIZE + WRITE_SIZE); rocPRIM's sorts and scans are not gsr:: kernels "
                              "and are left out (~0.2 GB)"}
json.dump(out, open(dst + "_pmc.json", "w"), indent=1, sort_keys=True)


def pmc_set(dirs, prefix, command, fetch_dir, marker=None, levels=4):
    """Per-kernel per-launch counter averages of one group of --pmc passes (the same layout as <dst>_pmc.json).  marker: a kernel
    the profiled script launches in front of every level of its schedule -- the rows are then keyed "L<k> <kernel>" with k = the
    number of markers seen so far modulo `levels` (dispatch order), one entry per level instead of an average over all of them."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    nl = collections.defaultdict(int)
    for d in dirs:
        for fn in newest(os.path.join(src, d, "*", "*counter_collection.csv")):
            seen_ = set()
            rows_ = sorted(csv.DictReader(open(fn)), key=lambda r: int(r["Dispatch_Id"]))
            marks, last_mark = 0, None
            for r in rows_:
                k = short(r["Kernel_Name"])
                if not k.startswith("gsr::"):
                    continue
                if marker:
                    if k.startswith(marker):
                        if r["Dispatch_Id"] != last_mark:
                            marks += 1; last_mark = r["Dispatch_Id"]
                        continue
                    if marks == 0:
                        continue                       # (the HEM levels in front of the schedule)
                    k = "L%d %s" % ((marks - 1) % levels, k)
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                key = (k, r["Dispatch_Id"])
                if d == fetch_dir and key not in seen_:
                    seen_.add(key); nl[k] += 1
    if not acc:
        return None
    o = {"kernel_build": bench.kernel_build_id(), "command": command,
         "note": "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a streaming read, x1.0 of random 64-byte "
                 "records, x0.67 of random 192-byte rows (profiles/archive/r03_fetch_calibration.txt): raw and x2 figures are both given"}
    for k, v in acc.items():
        n_ = max(1, nl.get(k, 1))
        e_ = {"launches_in_pass": n_}
        for c_, val in v.items():
            e_[c_ + "_per_launch"] = val / n_
        if "FETCH_SIZE" in v:
            e_["hbm_read_bytes_per_launch_raw"] = v["FETCH_SIZE"] * 1024 / n_
            e_["hbm_read_bytes_per_launch_x2_corrected"] = v["FETCH_SIZE"] * 1024 * 2 / n_
        if "WRITE_SIZE" in v:
            e_["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024 / n_
        if "TCC_HIT_sum" in v:
            e_["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        derived(e_, v, n_)
        o[k] = e_
    json.dump(o, open(dst + prefix, "w"), indent=1, sort_keys=True)
    return dst + prefix


pmc_set(("pmc_icp_fetch", "pmc_icp_write", "pmc_icp_l2", "pmc_icp_sq", "pmc_icp_sq2"), "_pmc_icp.json",
        "scripts/prof_icp.py 5000000 2  (the bench's 4-level point-to-plane schedule on the HEM levels of bench.py's own 5 M pair: 5 degrees / "
        "0.05 h; one entry per LEVEL and kernel -- L0 = 185 k points ... L3 = 5 M --, per-launch averages within the level)", "pmc_icp_fetch",
        marker="gsr::k_debug_logf")
pmc_set(("pmc_aniso_fetch", "pmc_aniso_write", "pmc_aniso_l2", "pmc_aniso_sq", "pmc_aniso_sq2"), "_pmc_aniso.json",
        "scripts/prof_hem.py 5000000 1 2 aniso  (level 1 of a 5 M surfel-shaped cloud)", "pmc_aniso_fetch")
f = newest(os.path.join(src, "icp_stats", "*", "*kernel_stats.csv"))
if f:
    rows = list(csv.DictReader(open(f[0])))
    with open(dst + "_icp_kernel_stats.csv", "w") as o:
        o.write("kernel,calls,total_ms,avg_us,pct\n")
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            o.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['TotalDurationNs'])/int(r['Calls'])/1e3:.2f},{100*float(r['TotalDurationNs'])/tot:.2f}\n")
for name in ("bench.json", "bench_prof.json", "bench_c2.json", "prof_icp.log", "pytest_gpu.log", "smoke.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, dst + "_" + name)
print("wrote", dst + "_kernel_stats.csv", dst + "_pmc.json")
