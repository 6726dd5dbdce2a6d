"""Per-kernel SQ counter averages from rocprofv3 --pmc counter_collection CSVs (one or more passes of the same program):
    python tools/dev_attn_counters.py <csv> [<csv> ...] [--filter attn]
Prints, per kernel, the mean of every counter per launch and a few derived ratios."""
import csv
import sys
from collections import defaultdict

args = sys.argv[1:]
flt = "attn"
if "--filter" in args:
    i = args.index("--filter")
    flt = args[i + 1]
    del args[i:i + 2]
paths = args
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for p in paths:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k, cs in acc.items():
    m = {c: v[0] / max(1, v[1]) for c, v in cs.items()}
    print(k[:90])
    print("   " + "  ".join(f"{c}={v:.3g}" for c, v in sorted(m.items())))
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        simd_cycles = g / 8 * 1024  # cycles x SIMDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_LDS_BANK_CONFLICT",
                  "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"):
            if c in m:
                print(f"   {c} / (cycles x SIMDs) = {m[c] / simd_cycles:.3f}   (x4 if the counter is in quad-cycles: {4 * m[c] / simd_cycles:.3f})")
