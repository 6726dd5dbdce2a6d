#!/usr/bin/env python3
"""First thing to run on a multi-GPU lease: does RCCL move the gradient all-reduce
over ALL xGMI links of the 8-GPU mesh, or over one ring?

    python tools/rccl_probe.py --gpus 8            # spawns the ranks itself (one process per GPU)

Per rank 0 it prints (1) the algorithm / protocol / channel lines RCCL logs at communicator creation
(NCCL_DEBUG=INFO, NCCL_DEBUG_SUBSYS=INIT,GRAPH,COLL), (2) the time and bus bandwidth of fp32 all-reduces of one
parameter bucket of each model size (XLS-R-300M layer = 50 MB, XLS-R-2B layer = 180 MB, the 2B front bucket = 560 MB) and
of 4.3 / 8.6 GB (a whole 2B gradient, bf16 / fp32 wire), (3) the same with the bucket sizes issued back to back on a side
stream while a GEMM loop runs on the main stream (the overlap the trainer relies on).

Reading the result: bus bandwidth = 2 (N-1)/N x bytes / time.  One ring over point-to-point xGMI is bound by ONE link,
~153 GB/s per direction: ~130-150 GB/s bus bandwidth.  All seven links in use shows as >~ 500 GB/s at 8 GPUs.  The step
budget is 8.64 GB fp32 per optimiser step against ~55 ms of backward: at 140 GB/s that is 108 ms (exposed),
at 600 GB/s 25 ms (hidden).  If the single-ring regime shows up, set the environment this script prints under "try next"
(more channels / direct mode) and re-run; the bench reads the same variables from the environment.
"""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def worker(args):
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    sizes_mb = [50, 180, 560, 4320, 8640]
    if rank == 0:
        print(f"world {world}; env: " + " ".join(f"{k}={v}" for k, v in sorted(os.environ.items())
                                                if k.startswith(("NCCL_", "RCCL_", "HSA_"))), flush=True)
    buf = torch.empty(max(sizes_mb) * 250_000, dtype=torch.float32, device=dev).normal_()
    comm = torch.cuda.Stream(device=dev)
    for mb in sizes_mb:
        n = mb * 250_000
        for _ in range(2):
            dist.all_reduce(buf[:n])
        torch.cuda.synchronize()
        dist.barrier()
        iters = 10 if mb < 1000 else 3
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_reduce(buf[:n])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        if rank == 0:
            bus = 2 * (world - 1) / world * n * 4 / dt / 1e9
            print(f"all_reduce fp32 {mb:5d} MB: {dt * 1e3:8.2f} ms  bus bandwidth {bus:7.1f} GB/s", flush=True)
    # overlap: 48 buckets of 180 MB on a side stream while bf16 GEMMs run on the main stream
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)

    def gemms(k):
        for _ in range(k):
            torch.mm(a, b)

    gemms(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gemms(60)
    torch.cuda.synchronize()
    t_gemm = time.perf_counter() - t0
    n = 180 * 250_000
    dist.barrier()
    t0 = time.perf_counter()
    comm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(comm):
        hs = [dist.all_reduce(buf[:n], async_op=True) for _ in range(48)]
    gemms(60)
    for h in hs:
        h.wait()
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()
    t_both = time.perf_counter() - t0
    if rank == 0:
        print(f"60 GEMMs alone {t_gemm * 1e3:.1f} ms; with 48 x 180 MB all-reduces on a side stream {t_both * 1e3:.1f} ms "
              f"(exposed communication {max(0.0, t_both - t_gemm) * 1e3:.1f} ms)", flush=True)
        print("try next if the bus bandwidth is single-link class: NCCL_MIN_NCHANNELS=28 NCCL_MAX_NCHANNELS=56 "
              "(four channels per xGMI link), RCCL_ENABLE_DIRECT=1 / NCCL_ALGO=Tree,Ring; compare the numbers above", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    args = ap.parse_args()
    if "WORLD_SIZE" in os.environ:
        return worker(args)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "INFO")
    env.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,COLL")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--local-addr", "127.0.0.1", str(Path(__file__).resolve()), "--gpus", str(args.gpus)]
    # RCCL's INFO lines go to stdout of every rank: keep the lines that say which algorithm / channels were chosen
    p = subprocess.run(cmd, env=env, capture_output=True, text=True)
    keep = ("Channel", "Ring", "Tree", "Algo", "algo", "channels", "nChannels", "XGMI", "xgmi", "P2P", "all_reduce", "GEMMs",
            "try next", "world ", "Connected", "threadThresholds", "comm ")
    seen = set()
    for line in (p.stdout + p.stderr).splitlines():
        if any(k in line for k in keep):
            key = line.split("] ", 1)[-1][:120]
            if key not in seen:
                seen.add(key)
                print(line[:220])
    raise SystemExit(p.returncode)


if __name__ == "__main__":
    main()
