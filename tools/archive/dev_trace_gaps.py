"""Idle time of the GPU inside the steady-state steps of a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --no-fwd-bwd
    python tools/dev_trace_gaps.py /tmp/kt/.../*_kernel_trace.csv [adamw_kernel]
Steps are delimited by the launches of the marker kernel (default: adamw_kernel, one per step).  Prints per step: wall,
union of busy intervals, idle, number of launches, and the histogram of the gaps between consecutive busy intervals."""
import csv
import sys
from collections import Counter

path = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
print(f"{len(rows)} launches, {len(marks)} markers")
for a, b in zip(marks[-4:-1], marks[-3:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], rows[b][0]
    busy, gaps = 0, []
    cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    busy += min(cur_e, t1) - cur_s
    ssum = sum(e - s for s, e, _ in seg)
    h = Counter(min(int(g / 500), 20) for g in gaps)
    print(f"step: wall {(t1 - t0) / 1e6:.2f} ms, busy (union) {busy / 1e6:.2f}, idle {(t1 - t0 - busy) / 1e6:.2f}, "
          f"sum of durations {ssum / 1e6:.2f}, launches {len(seg)}, gaps {len(gaps)} (mean {sum(gaps) / max(1, len(gaps)) / 1e3:.2f} us)")
    print("   gap histogram (0.5-us bins):", " ".join(f"{k * 0.5:.1f}:{v}" for k, v in sorted(h.items())))
    big = sorted(((g, i) for i, g in enumerate(gaps)), reverse=True)[:5]
    print("   largest gaps (us):", [round(g / 1e3, 1) for g, _ in big])
