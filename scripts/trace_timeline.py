"""Timeline of the last `run_level` of a rocprofv3 --kernel-trace csv: kernels, durations, idle gaps between them."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_prep"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # which occurrence of `first` starts the window (-1 = last)
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[nth]
i1 = starts[nth + 1] if nth != -1 and nth + 1 < len(starts) else len(rows)
if len(sys.argv) > 4 and sys.argv[4] == "end":                # the window runs to the end of the trace (e.g. the last four registrations of a schedule)
    i1 = len(rows)
win = rows[i0:i1]
t0 = int(win[0]["Start_Timestamp"]); prev = None; busy = 0; idle = 0; big = []
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    if prev and s > prev: idle += s - prev
    busy += e - max(s, prev or s)
    if gap > 3.0: big.append((gap, r["Kernel_Name"][:50]))
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:9.1f}  gap {gap:7.1f}  {r['Kernel_Name'][:70]}")
    prev = max(prev or e, e)
print(f"window {(prev - t0) / 1e3:.1f} us, idle {idle / 1e3:.1f} us in gaps; gaps > 3 us: {len(big)} totalling {sum(g for g, _ in big):.1f} us")
for g, k in big: print(f"   {g:7.1f} us before {k}")
