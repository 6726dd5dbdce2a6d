"""Per-launch view of tools/r05_hog_gemm.py: each GEMM launch shape of an XLS-R-2B layer alone, beside N hog workgroups
(ca_debug_cu_hog) with the N = 1 settings, and beside them with ca_gemm_set_compute_cus(256 - N):
   python tools/r05_hog_shapes.py [--hog 16] [--lds 98304]"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hog", type=int, default=16)
ap.add_argument("--lds", type=int, default=96 * 1024)
ap.add_argument("--threads", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
src = Path(__file__).with_name("r05_gemm_table.py").read_text().split('print(f"# {torch.cuda.get_device_name(0)}')[0]
src = src.replace("args = ap.parse_args()", "args = ap.parse_args([])")
ns = {"__file__": str(Path(__file__).with_name("r05_gemm_table.py")), "__name__": "shapes"}
exec(compile(src, "r05_gemm_table.py", "exec"), ns)
lib = ops.lib()
side = torch.cuda.Stream()
ncu = torch.cuda.get_device_properties(0).multi_processor_count


def timed(fn):
    main = torch.cuda.current_stream()
    for _ in range(3):
        fn()
    main.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    main.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


def hogged(fn, cus):
    lib.ca_gemm_set_compute_cus(cus)
    with torch.cuda.stream(side):
        ops.check(lib.ca_debug_cu_hog(args.hog, args.threads, args.lds, 200.0, side.cuda_stream), "ca_debug_cu_hog")
    time.sleep(0.01)
    t = timed(fn)
    side.synchronize()
    lib.ca_gemm_set_compute_cus(0)
    return t


print(f"# {args.hog} hog workgroups x {args.threads} threads x {args.lds} B LDS; us per launch; ideal = alone x {ncu} / {ncu - args.hog}")
print(f"{'launch':42s} {'alone':>8s} {'ideal':>8s} {'hog':>8s} {'hog,dyn':>8s} {'hog,cap':>8s}")
for name, M, N, K, al, bl, kind in ns["layer_shapes"](3992, 1920, 7680):
    fn = ns["build"](M, N, K, al, bl, kind)
    alone = timed(fn)
    a, b, c = hogged(fn, 0), hogged(fn, ncu), hogged(fn, ncu - args.hog)
    print(f"{name:42s} {alone:8.1f} {alone * ncu / (ncu - args.hog):8.1f} {a:8.1f} {b:8.1f} {c:8.1f}", flush=True)
    del fn
    torch.cuda.empty_cache()
