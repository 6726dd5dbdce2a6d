"""Tuning aid for the kernel S / kernel X dispatch rule: forward and data-gradient shapes of the supported models,
timed with each kernel forced and with the automatic choice.
usage: python tools/dev_gemm_rule.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=12):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = []
for name, M, d, f in (("xlsr-2b", 3992, 1920, 7680), ("xlsr-300m", 3992, 1024, 4096), ("xlsr-1b", 3992, 1280, 5120), ("whisper-medium", 12000, 1024, 4096),
                      ("whisper-turbo", 12000, 1280, 5120)):
    for bl in (0, 1):
        shapes += [(name, M, d, d, bl), (name, M, 3 * d, d, bl), (name, M, d, 3 * d, bl), (name, M, f, d, bl), (name, M, d, f, bl)]
for name, M, N, K, bl in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    Bt = B.t().contiguous() if bl else B
    Cd = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(M=M, N=N, K=K, a_layout=0, b_layout=bl, lda=K, ldb=N if bl else K, ldc=N)
    t = {}
    for force in (1, 2, 3, 0):
        ops.lib().ca_gemm_force_kernel(force)
        t[force] = timeit(lambda: ops.gemm(A, Bt, Cd, **kw))
    ops.lib().ca_gemm_force_kernel(0)
    names = {1: "S", 2: "L", 3: "X"}
    best = min((1, 2, 3), key=lambda k: t[k])
    auto = min((1, 2, 3), key=lambda k: abs(t[0] - t[k]))
    flag = "" if best == auto or (t[auto] - t[best]) / t[best] < 0.03 else "   <-- rule picks a slower one"
    xt = ((M + 255) // 256) * ((N + 255) // 256)
    tf = 2.0 * M * N * K / t[0] / 1e6
    print(f"{name:15s} M{M:6d} N{N:5d} K{K:5d} {'NT' if not bl else 'NN'}  S {t[1]:7.1f}  L {t[2]:7.1f}  X {t[3]:7.1f}  auto {t[0]:7.1f} "
          f"({names[auto]}, {tf:6.0f} TF) xtiles {xt:4d}{flag}")
