"""Kernels ranked by LDS bank-conflict cycles from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE pass:
    python tools/dev_lds_conflicts.py <counter_collection.csv>"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[r["Kernel_Name"]] += 1
rows = []
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    if g <= 0:
        continue
    simd = g / 8 * 1024
    rows.append((c.get("SQ_LDS_BANK_CONFLICT", 0.0), k, n[k], c.get("SQ_LDS_BANK_CONFLICT", 0.0) / simd, c.get("SQ_LDS_IDX_ACTIVE", 0.0) / simd, g))
rows.sort(reverse=True)
tot = sum(r[5] for r in rows)
for cf, k, cnt, fc, fa, g in rows[:18]:
    print(f"{k[:70]:70s} launches {cnt:5d}  share of GPU cycles {g / tot:5.3f}  conflict {fc:6.3f}  LDS active {fa:6.3f}")
