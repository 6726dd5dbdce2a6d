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
cmd = sys.argv[3] if len(sys.argv) > 3 else "python bench.py --steps 4 --warmup 2 --no-cpu-baseline"


def one(pattern):
    f = glob.glob(str(ROOT / "gpurun_out" / pattern))
    return max(f, key=lambda p: Path(p).stat().st_mtime) if f else None


stats = one(f"prof_{tag}_stats/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}",
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
# flat view used by bench.py for the default workload
if tag == "r01":
    old.update(traffic)
out.write_text(json.dumps(old, indent=1, sort_keys=True))
print("step ms", tot / 1e6 / nsteps)
for r in rows[:14]:
    print(f"  {r['Name'][:64]:64s} calls/step {int(r['Calls'])/nsteps:7.1f} ms/step {float(r['TotalDurationNs'])/1e6/nsteps:8.2f} avg {float(r['AverageNs'])/1e3:8.1f} us "
          f"traffic/launch {traffic.get(r['Name'], 0)/1e6:8.1f} MB")
