"""Print the dispatch timeline of the LAST HEM level found in a rocprofv3 kernel trace (csv): kernel, duration, and
the idle gap before it -- shows host round trips and launch gaps as well as kernel time.
usage: python scripts/trace_level.py <dir with *kernel_trace.csv> [which_level_from_end=1]"""
import csv, glob, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_prep" in r["Kernel_Name"]]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
tot_k = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("gsr::", "")
    if name.startswith("void rocprim") or name.startswith("rocprim"):
        name = "rocprim:" + name.split("<")[0].split("::")[-1] + ("<" + name.split("<", 2)[1][:40] if "<" in name else "")
    tot_k += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name[:90]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot_k / 1e3:.1f} us, dispatches {i1 - i0}")
