"""gpurun_out/prof_<tag>_steady{2,6} (tools/profile_steady.sh) -> profiles/<tag>_per_step.csv: per kernel the launches
and the time of ONE steady-state step = (6-step run - 2-step run) / 4.  Set-up kernels (initialisation fills, warm-up)
appear in both runs and cancel; a kernel with zero launches per step is listed in the trailer as set-up only."""
import csv
import glob
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
cmd = sys.argv[2] if len(sys.argv) > 2 else "python bench.py --steps {2,6} --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd"


def load(n):
    f = glob.glob(str(ROOT / "gpurun_out" / f"prof_{tag}_steady{n}" / "*" / "*kernel_stats.csv"))
    f = max(f, key=lambda p: Path(p).stat().st_mtime)
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


a, b = load(2), load(6)
rows, setup = [], []
for name, (cb, tb) in b.items():
    ca, ta = a.get(name, (0, 0.0))
    calls, ns = (cb - ca) / 4.0, (tb - ta) / 4.0
    if calls <= 0:
        setup.append((name, cb))
        continue
    rows.append((name, calls, ns))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
lines = [f"# CA_WGRAD_STREAM=0 CA_OPT_OVERLAP=0 rocprofv3 --kernel-trace --stats -- {cmd}",
         "# per-step figures = (run with 6 timed steps - run with 2 timed steps) / 4: set-up and warm-up cancel",
         f"# 1x MI355X; one steady-state step = {sum(r[1] for r in rows):.0f} kernel launches, {tot / 1e6:.2f} ms of kernel time (serialised)",
         "Name,CallsPerStep,NsPerStep,AverageNs,Percentage"]
for name, calls, ns in rows:
    lines.append(f'"{name}",{calls:g},{ns:.0f},{ns / calls:.0f},{100 * ns / tot:.2f}')
lines.append("# launched during set-up / warm-up only (no launch in a steady-state step): " +
             "; ".join(f"{n.split('(')[0][-60:]} x{c}" for n, c in setup[:40]))
(ROOT / "profiles" / f"{tag}_per_step.csv").write_text("\n".join(lines) + "\n")
print("\n".join(lines[:28]))
