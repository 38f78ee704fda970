"""Per-kernel durations and the gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (last N dispatches)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sub = sys.argv[2] if len(sys.argv) > 2 else "icp"
rows = [r for r in rows if sub in r["Kernel_Name"]]
last = rows[-int(sys.argv[3]) if len(sys.argv) > 3 else -40:]
prev = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:40]:40s} dur {(e - s) / 1e3:8.2f} us  gap {((s - prev) / 1e3 if prev else 0):8.2f} us  grid {r.get('Grid_Size_X', '?')} wg {r.get('Workgroup_Size_X', '?')}")
    prev = e
