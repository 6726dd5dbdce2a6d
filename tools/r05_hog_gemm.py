"""GEMMs beside a resident collective, emulated on one GPU (round-5 review item 2):
   python tools/r05_hog_gemm.py [--layers 4] [--hogs 16,32,64]
A test-only kernel (`ca_debug_cu_hog`: N idle workgroups, one per CU, holding 96 KiB of LDS each) sits on a side stream
for the whole measurement, the way an RCCL ring kernel occupies CUs during the backward of an N > 1 run.  Beside it the
GEMM launches of XLS-R-2B transformer layers (forward, data gradients, weight gradients: tools/r05_gemm_table.py's
shapes with their epilogues) are timed in three settings:
   static   the N = 1 default: persistent launches of one workgroup per CU, first tile of a workgroup static
   dynamic  ca_gemm_set_compute_cus(256): every tile from the counter (a workgroup that starts late exits)
   capped   ca_gemm_set_compute_cus(256 - N): also the launch size and the tile-shape rule follow the CUs that are left
Ideal = the un-hogged time x 256 / (256 - N)."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--hogs", default="16,32,64")
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = "cuda:0"
sys.argv = [sys.argv[0]]
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("r05_gemm_table_shapes", Path(__file__).with_name("r05_gemm_table.py"))
# (only the shape list and the launch builders of the table tool are needed: import them without running its main part)
src = Path(__file__).with_name("r05_gemm_table.py").read_text().split('print(f"# {torch.cuda.get_device_name(0)}')[0]
src = src.replace("args = ap.parse_args()", "args = ap.parse_args([])")
ns = {"__file__": str(Path(__file__).with_name("r05_gemm_table.py")), "__name__": "shapes"}
exec(compile(src, "r05_gemm_table.py", "exec"), ns)
shapes = ns["layer_shapes"](3992, 1920, 7680)
groups = {"forward": shapes[0:4], "data gradients": shapes[4:8], "weight gradients": shapes[8:12]}
fns = {g: [ns["build"](M, N, K, al, bl, kind) for _, M, N, K, al, bl, kind in lst] for g, lst in groups.items()}
lib = ops.lib()
side = torch.cuda.Stream()
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def chain(which):
    for _ in range(args.layers):
        for g in which:
            for fn in fns[g]:
                fn()


def timed(which):
    main = torch.cuda.current_stream()
    chain(which)  # warm-up
    main.synchronize()  # (this stream only: a device-wide synchronise would wait for the hog to leave)
    best = 1e9
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chain(which)
        e1.record()
        main.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / args.layers  # us per layer


def with_hog(n, which, ms=600.0):
    """The same timing with n hog workgroups resident: started first, given a moment to occupy their CUs."""
    with torch.cuda.stream(side):
        ops.check(lib.ca_debug_cu_hog(n, 256, 96 * 1024, ms, side.cuda_stream), "ca_debug_cu_hog")
    time.sleep(0.02)
    t0 = time.time()
    t = timed(which)
    took = time.time() - t0
    assert took < ms * 1e-3 * 0.9, f"the hog ({ms} ms) left before the measurement ended ({took * 1e3:.0f} ms)"
    side.synchronize()
    return t


print(f"# {torch.cuda.get_device_name(0)}, {ncu} CUs; XLS-R-2B layer launches (M = 3992, d = 1920, ffn = 7680), {args.layers} layers per "
      f"measurement, best of {args.iters}; us per layer")
print(f"{'launch group':18s} {'hog CUs':>7s} {'alone':>8s} {'ideal':>8s} {'static':>8s} {'dynamic':>8s} {'capped':>8s}   slowdown / ideal: static dynamic capped")
for gname in ("forward", "data gradients", "weight gradients", "all"):
    which = list(groups) if gname == "all" else [gname]
    lib.ca_gemm_set_compute_cus(0)
    alone = timed(which)
    for n in [int(x) for x in args.hogs.split(",")]:
        ideal = alone * ncu / (ncu - n)
        res = []
        for setting in (0, ncu, ncu - n):
            lib.ca_gemm_set_compute_cus(setting)
            res.append(with_hog(n, which))
        lib.ca_gemm_set_compute_cus(0)
        print(f"{gname:18s} {n:7d} {alone:8.1f} {ideal:8.1f} {res[0]:8.1f} {res[1]:8.1f} {res[2]:8.1f}   "
              + " ".join(f"{r / ideal:6.3f}" for r in res), flush=True)
