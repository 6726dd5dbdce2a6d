"""MFMA-pipe busy share per launch shape: summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` run of
`tools/r05_gemm_table.py --kernels 0 --iters 3` (every shape = 6 consecutive launches of one GEMM kernel):
   python tools/r05_gemm_pmc.py <counter_collection.csv> <table output>
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (MI355X_MICROARCH.md, cycle constants)."""
import collections
import csv
import sys

rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "ca_gemm" not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].split("(")[0].replace("void ", "")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = [rows[k] for k in sorted(rows)]
import re

shapes = [l for l in open(sys.argv[2]) if re.search(r"\s(NT|NN|TN|TT)\s+\d", l) and not l.startswith(("launch", "W20", "E20", "I20"))]
per = 6
print(f"# {len(disp)} GEMM dispatches, {len(shapes)} shapes")
for i, line in enumerate(shapes):
    grp = disp[i * per:(i + 1) * per]
    if not grp:
        break
    busy = sum(g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for g in grp)
    act = sum(g.get("GRBM_GUI_ACTIVE", 0.0) for g in grp)
    frac = busy / (act / 8.0 * 1024.0) if act else 0.0
    print(f"{line.rstrip()[:110]:110s}  {grp[0]['k']:34s} mfma_busy {frac:5.3f}")
