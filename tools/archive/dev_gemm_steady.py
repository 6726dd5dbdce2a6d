"""Steady-state rate of kernel X per CU: one round of 256 tiles with a long K, (a) on a real 4096 x 4096 problem,
(b) with every tile reading the SAME operand panels (batch stride 0: everything L2-resident), per operand form.
usage: python tools/dev_gemm_steady.py [K]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ops.lib().ca_gemm_force_kernel(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for al, bl in ((0, 0), (0, 1), (1, 0), (1, 1)):
    M = N = 4096
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    At = A.t().contiguous() if al else A
    Bt = B.t().contiguous() if bl else B
    Cd = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(At, Bt, Cd, M=M, N=N, K=K, a_layout=al, b_layout=bl, lda=M if al else K,
                                ldb=N if bl else K, ldc=N))
    # same panels for all 256 tiles
    m = 256
    a = (A[:m].t().contiguous() if al else A[:m].contiguous())
    b = (B[:m].t().contiguous() if bl else B[:m].contiguous())
    c = torch.zeros(256, m, m, dtype=torch.bfloat16, device=dev)
    t2 = timeit(lambda: ops.gemm(a, b, c, M=m, N=m, K=K, a_layout=al, b_layout=bl, lda=m if al else K,
                                 ldb=m if bl else K, ldc=m, batch1=256, sA=(0, 0), sB=(0, 0), sC=(m * m, 0)))
    fl = 2 * M * N * K / 1e9
    print(f"al{al} bl{bl} K{K}: real {t*1e3:7.1f} us {fl/t:7.1f} TF ({t*1e6/(K/64):.0f} ns/K-step) | "
          f"shared panels {t2*1e3:7.1f} us {fl/t2:7.1f} TF ({t2*1e6/(K/64):.0f} ns/K-step)")
