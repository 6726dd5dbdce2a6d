"""Kernel time by (kernel, grid size) from a rocprofv3 kernel trace (CSV), steady part only (the last `frac` of the launches):
    python tools/dev_trace_by_grid.py <kernel_trace.csv> [frac=0.5] [name filter]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    if flt and flt not in r["Kernel_Name"]:
        continue
    k = (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    a = agg[k]
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"{len(rows)} launches, {tot / 1e3:.2f} ms of kernel time")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{a[1] / tot * 100:5.1f}%  {a[0]:6d} x {a[1] / a[0]:8.2f} us  grid {k[1]:>8s} wg {k[2]:>4s}  {k[0]}")
