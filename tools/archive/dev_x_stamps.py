"""Where a kernel-X workgroup's cycles go (needs the -DX_STAMPS build of gemm.hip: CORAL_AMD_LIB=coral_amd/libvariant_stamps.so)."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
lib = ops.lib()
raw = ctypes.CDLL(str(Path(__import__("os").environ["CORAL_AMD_LIB"]).resolve()))


def stamps():
    n = 512 * 8 * 8
    buf = (ctypes.c_longlong * n)()
    raw.ca_gemm_x_stamps(buf, n)
    return np.frombuffer(buf, dtype=np.int64).reshape(512, 8, 8).copy()


def report(name, nwg):
    s = stamps()[:nwg].astype(np.float64)
    pro, loop, epi, dma, bar, nk, rt = [s[:, :, i] for i in range(7)]
    nk = nk.mean()
    ok = rt > 0
    print(f"      shader clock inside the main loop (d s_memtime / d s_memrealtime x 100 MHz, median over waves): "
          f"{np.median(loop[ok] / rt[ok]) * 0.1:.2f} GHz")
    print(f"{name}: nk {nk:.0f} | prologue {pro.mean():7.0f} | loop {loop.mean():8.0f} = {loop.mean() / nk:6.0f}/k-step "
          f"(2048 = MFMA floor at 2 waves per SIMD) | dma wait {dma.mean() / nk:5.0f}/k-step  barrier wait {bar.mean() / nk:5.0f}/k-step "
          f"| epilogue {epi.mean():7.0f}   [s_memtime ticks]")
    print(f"      per-wave loop min {loop.min():.0f} max {loop.max():.0f}; dma wait by wave {np.round(dma.mean(0) / nk).tolist()}; "
          f"barrier wait by wave {np.round(bar.mean(0) / nk).tolist()}")


def run(name, M, N, K, al, bl, out_f32, **extra):
    A = torch.randn((K, M) if al else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if bl else (N, K), device=dev).to(torch.bfloat16)
    Cd = torch.zeros(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    kw = dict(M=M, N=N, K=K, a_layout=al, b_layout=bl, lda=M if al else K, ldb=N if bl else K, ldc=N, out_f32=out_f32, **extra)
    lib.ca_gemm_force_kernel(3)
    for _ in range(5):
        ops.gemm(A, B, Cd, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(A, B, Cd, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{name}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
    report(name, min(256, ((M + 255) // 256) * ((N + 255) // 256)))
    lib.ca_gemm_force_kernel(0)


if len(sys.argv) > 1:  # M N K al bl [out_f32]
    a = [int(x) for x in sys.argv[1:]]
    run(" ".join(sys.argv[1:]), a[0], a[1], a[2], a[3], a[4], bool(a[5]) if len(a) > 5 else False)
    sys.exit(0)
run("wgrad fc (TN, fp32 out)", 7680, 1920, 3992, 1, 1, True)
run("fc1 fwd plain (NT)", 3992, 7680, 1920, 0, 0, False)
run("fc2 dgrad plain (NN)", 3992, 7680, 1920, 0, 1, False)
run("big square (NT) 8192^3", 8192, 8192, 8192, 0, 0, False)
