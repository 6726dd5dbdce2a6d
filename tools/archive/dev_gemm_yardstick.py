"""Development yardstick (never part of the product path): our GEMM against torch.matmul (hipBLASLt/rocBLAS) on the
same box and the same random operands, for the step's shapes and a large square.
usage: python tools/dev_gemm_yardstick.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (M, N, K, al, bl): al/bl = 1 means the operand is stored with M (N) contiguous
shapes = [
    (8192, 8192, 8192, 0, 0), (8192, 8192, 8192, 0, 1), (8192, 8192, 8192, 1, 1),
    (3992, 1920, 1920, 0, 0), (3992, 5760, 1920, 0, 0), (3992, 7680, 1920, 0, 0), (3992, 1920, 7680, 0, 0),
    (3992, 1920, 1920, 0, 1), (3992, 1920, 5760, 0, 1), (3992, 7680, 1920, 0, 1), (3992, 1920, 7680, 0, 1),
    (1920, 7680, 3992, 1, 1), (7680, 1920, 3992, 1, 1), (1920, 1920, 3992, 1, 1), (5760, 1920, 3992, 1, 1),
]
for M, N, K, al, bl in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    At = A.t().contiguous() if al else A
    Bt = B.t().contiguous() if bl else B
    Cd = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(M=M, N=N, K=K, a_layout=al, b_layout=bl, lda=M if al else K, ldb=N if bl else K, ldc=N)
    ours = timeit(lambda: ops.gemm(At, Bt, Cd, **kw))
    a_op = At.t() if al else At  # logical (M, K)
    b_op = Bt if bl else Bt.t()  # logical (K, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib = timeit(lambda: torch.matmul(a_op, b_op, out=out))
    err = (out.float() - Cd.float()).abs().max().item()
    fl = 2 * M * N * K / 1e9
    print(f"M{M} N{N} K{K} al{al} bl{bl}: ours {ours*1e3:7.1f} us {fl/ours:7.1f} TF | hipBLASLt {lib*1e3:7.1f} us {fl/lib:7.1f} TF | maxdiff {err:.3g}")
