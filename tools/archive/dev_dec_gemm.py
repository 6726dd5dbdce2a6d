"""Decoder-side GEMMs of the Whisper training step (teacher-forced, M = B x L <= 960 rows): forward (NT), data gradient
(NN) and weight gradient (TT, fp32 out) per kernel choice.  usage: python tools/dev_dec_gemm.py [M]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
NAMES = {0: "auto", 1: "S", 2: "L", 3: "X", 5: "M"}
M = int(sys.argv[1]) if len(sys.argv) > 1 else 904


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for d, f in ((1024, 4096), (1280, 5120)):
    for N, K in ((3 * d, d), (d, d), (f, d), (d, f)):
        X = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = torch.randn(N, K, device=dev).to(torch.bfloat16)
        Y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        dX = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
        dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
        row = f"M{M} N{N:5d} K{K:5d}: "
        for force in (0, 1, 2, 5):
            ops.lib().ca_gemm_force_kernel(force)
            tf = timeit(lambda: ops.gemm(X, W, Y, M=M, N=N, K=K, a_layout=0, b_layout=0, lda=K, ldb=K, ldc=N))
            td = timeit(lambda: ops.gemm(Y, W, dX, M=M, N=K, K=N, a_layout=0, b_layout=1, lda=N, ldb=K, ldc=K))
            tw = timeit(lambda: ops.gemm(Y, X, dW, M=N, N=K, K=M, a_layout=1, b_layout=1, lda=N, ldb=K, ldc=K, out_f32=True))
            row += f" [{NAMES[force]}] fwd {tf:5.1f} dgrad {td:5.1f} wgrad {tw:5.1f} |"
        ops.lib().ca_gemm_force_kernel(0)
        print(row, f" weights {N * K * 2 / 1e6:.1f} MB, {2.0 * M * N * K / 1e9:.2f} GFLOP")
