"""whisper-large-turbo step, bf16 ('b') and fp8 ('f') runs INSIDE one process (what bench.py's also_turbo does once):
python tools/dev_turbo_inproc.py [bfbfbf] [sleep_s] [gc|arena]
With `--split <kernel_trace.csv> <n_runs>`: split a rocprofv3 kernel trace of such a process into n equal parts by launch
count and print the kernels whose total differs most between the first and last part."""
import gc
import sys
import time
import types
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def split(path, n):
    import collections
    import csv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // n
    parts = []
    for i in (0, n - 1):
        tot = collections.Counter()
        cnt = collections.Counter()
        for r in rows[i * per:(i + 1) * per]:
            k = r["Kernel_Name"][:70]
            tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[k] += 1
        parts.append((tot, cnt))
    (t0, c0), (t1, c1) = parts
    print(f"{len(rows)} launches, {per} per part; part totals {sum(t0.values())/1e3:.2f} ms vs {sum(t1.values())/1e3:.2f} ms")
    for k in sorted(set(t0) | set(t1), key=lambda k: -abs(t1[k] - t0[k]))[:25]:
        print(f"{t1[k]-t0[k]:+10.1f} us  {t0[k]:10.1f} ({c0[k]:5d}) -> {t1[k]:10.1f} ({c1[k]:5d})  {k}")


if len(sys.argv) > 1 and sys.argv[1] == "--split":
    split(sys.argv[2], int(sys.argv[3]))
    sys.exit(0)

import torch  # noqa: E402

import bench  # noqa: E402

seq = sys.argv[1] if len(sys.argv) > 1 else "bfbfbf"
nap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
collect = len(sys.argv) > 3 and sys.argv[3] == "gc"
args = types.SimpleNamespace(batch=8, steps=4, warmup=2, grad_wire="fp32", zero_stage=0, decode_tokens=32)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
arena = len(sys.argv) > 3 and sys.argv[3] == "arena"
if arena:  # one big segment for the caching allocator to carve everything from; never returned to the driver
    x = torch.empty(100 << 30, dtype=torch.uint8, device=dev)
    del x
for c in seq:
    r = bench.whisper_measure("whisper-large-turbo", args, 1, 0, dev, decode=False, fp8=(c == "f"), B=8, steps=4, warmup=2)
    print("fp8 " if c == "f" else "bf16", round(r["ms_per_step"], 2), flush=True)
    del r
    if collect:
        gc.collect()
    if not arena:
        torch.cuda.empty_cache()
    print("   reserved after empty_cache: %.2f GiB" % (torch.cuda.memory_reserved() / 2**30), flush=True)
    time.sleep(nap)
