"""Print the last N dispatches of a rocprofv3 kernel trace (csv) with gaps -- e.g. the ICP iteration loop.
usage: python scripts/trace_tail.py <dir> [N=60] [name-filter]"""
import csv, glob, sys
d = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
flt = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if flt:
    idx = [i for i, r in enumerate(rows) if flt in r["Kernel_Name"]]
    rows = rows[idx[0] - 2: idx[0] + N] if idx else rows[-N:]
else:
    rows = rows[-N:]
prev = int(rows[0]["Start_Timestamp"])
t0 = prev
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'].replace('gsr::', '')[:80]}")
    prev = e
