"""Turn gpurun_out/prof_<tag>_{stats,fetch,write} (tools/profile_bench.sh) into committed files:
profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, verbatim columns) and
profiles/pmc_traffic.json (kernel symbol -> HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE KiB,
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd"


def one(pattern):
    f = glob.glob(str(ROOT / "gpurun_out" / pattern))
    return max(f, key=lambda p: Path(p).stat().st_mtime) if f else None


stats = one(f"prof_{tag}_stats/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = [f"# CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}",
         "# (kernels serialised on one stream for per-kernel durations; the timed bench overlaps the optimiser and the weight gradients on side streams)",
         f"# 1x MI355X; {nsteps} train steps in the run (warm-up + timed + 2 hipEvent-profiled); "
         f"total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / nsteps:.2f} ms/step",
         "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev"]
for r in rows:
    lines.append(",".join(['"' + r["Name"] + '"'] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]))
(ROOT / "profiles" / f"{tag}_kernel_stats.csv").write_text("\n".join(lines) + "\n")

traffic = {}
acc = {}
for kind in ("fetch", "write"):
    f = one(f"prof_{tag}_{kind}/*/*counter_collection.csv")
    if not f:
        continue
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    acc[kind] = {k: sum(v) / len(v) for k, v in d.items()}
for k in acc.get("fetch", {}):
    fetch_kib = acc["fetch"][k]
    write_kib = acc.get("write", {}).get(k, 0.0)
    traffic[k] = round((2.0 * fetch_kib + write_kib) * 1024.0)  # bytes per launch
out = ROOT / "profiles" / "pmc_traffic.json"
old = json.loads(out.read_text()) if out.exists() else {}
old[tag] = {"note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024; averaged over the launches of "
                    "one bench step pair (separate --pmc passes)", "kernels": traffic}
# flat view used by bench.py for the default workload: the newest round's numbers
if tag.startswith("r0") and not tag.endswith("small") and "whisper" not in tag:
    old.update(traffic)
out.write_text(json.dumps(old, indent=1, sort_keys=True))
print("step ms", tot / 1e6 / nsteps)
for r in rows[:14]:
    print(f"  {r['Name'][:64]:64s} calls/step {int(r['Calls'])/nsteps:7.1f} ms/step {float(r['TotalDurationNs'])/1e6/nsteps:8.2f} avg {float(r['AverageNs'])/1e3:8.1f} us "
          f"traffic/launch {traffic.get(r['Name'], 0)/1e6:8.1f} MB")

# ---- SQ counters per kernel (tools/profile_round.sh pass 2), if collected -------------------------------------------
sq = one(f"prof_{tag}_sq/*/*counter_collection.csv")
if sq:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(sq)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_WAIT_ANY",
             "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
    out_lines = ["# rocprofv3 --pmc " + " ".join(names) + " -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-fwd-bwd",
                 "# per-launch averages over the launches of the run; SQ_*_CYCLES of waves are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and",
                 "# GRBM_GUI_ACTIVE are cycles summed over the 8 XCDs (MI355X_MICROARCH.md, cycle constants); mfma_busy_frac =",
                 "# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = share of the kernel's cycles its matrix pipes were busy",
                 "Kernel,Launches," + ",".join(names) + ",valu_per_mfma,mfma_busy_frac"]
    rows_sq = []
    for k, d in agg.items():
        n = max(len(v) for v in d.values())
        avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else 0.0) for c in names}
        vpm = avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"] if avg["SQ_INSTS_MFMA"] else 0.0
        busy = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if avg["GRBM_GUI_ACTIVE"] else 0.0
        rows_sq.append((sum(d.get("SQ_BUSY_CYCLES", [0])), k, n, avg, vpm, busy))
    for _, k, n, avg, vpm, busy in sorted(rows_sq, reverse=True)[:40]:
        out_lines.append('"' + k + '",' + str(n) + "," + ",".join(f"{avg[c]:.0f}" for c in names) + f",{vpm:.2f},{busy:.3f}")
    (ROOT / "profiles" / f"{tag}_sq_counters.csv").write_text("\n".join(out_lines) + "\n")
    print("wrote", f"profiles/{tag}_sq_counters.csv")
dec = one(f"prof_{tag}_decode_stats/*/*kernel_stats.csv")
if dec:
    rows_d = list(csv.DictReader(open(dec)))
    totd = sum(float(r["TotalDurationNs"]) for r in rows_d)
    lines_d = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --model whisper-medium --decode --steps 4 --warmup 2",
               f"# 6 decode passes of 8 x 30 s clips, 32 new tokens each; total kernel time {totd / 1e6:.1f} ms",
               "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev"]
    for r in rows_d:
        lines_d.append(",".join(['"' + r["Name"] + '"'] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")]))
    (ROOT / "profiles" / f"{tag}_decode_kernel_stats.csv").write_text("\n".join(lines_d) + "\n")
    print("wrote", f"profiles/{tag}_decode_kernel_stats.csv")
