"""What the fused FFN epilogues cost on top of the plain GEMM of the same shape (XLS-R-2B, kernel X):  python tools/dev_epilogue_cost.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
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
    return e0.elapsed_time(e1) / iters * 1e3


M, d, f = 3992, 1920, 7680
x = torch.randn(M, d, device=dev).to(torch.bfloat16)
W1 = (0.02 * torch.randn(f, d, device=dev)).to(torch.bfloat16)
b1 = torch.zeros(f, device=dev)
u = torch.empty(M, f, dtype=torch.bfloat16, device=dev)
g = torch.empty(M, f, dtype=torch.bfloat16, device=dev)
R = torch.randn(M, f, device=dev).to(torch.bfloat16)
kw = dict(M=M, N=f, K=d, a_layout=0, b_layout=0, lda=d, ldb=d, ldc=f)
print("fc1 forward  [3992 x 7680 x 1920], NT")
print(f"  plain                         {timeit(lambda: ops.gemm(x, W1, u, **kw)):7.1f} us")
print(f"  + bias                        {timeit(lambda: ops.gemm(x, W1, u, bias=b1, **kw)):7.1f} us")
print(f"  + bias + residual read        {timeit(lambda: ops.gemm(x, W1, u, bias=b1, R=R, ldr=f, epilogue=ops.EPI_RESIDUAL, **kw)):7.1f} us")
print(f"  + bias + GELU, 2 outputs      {timeit(lambda: ops.gemm(x, W1, u, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, **kw)):7.1f} us")
print(f"  + bias + GELU + dropout, 2 out{timeit(lambda: ops.gemm(x, W1, u, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, dropout_p=0.1, dropout_seed=7, **kw)):7.1f} us")
print(f"  + bias + GELU + dropout, g only{timeit(lambda: ops.gemm(x, W1, None, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, dropout_p=0.1, dropout_seed=7, **kw)):6.1f} us")
dh = torch.randn(M, d, device=dev).to(torch.bfloat16)
W2 = (0.02 * torch.randn(d, f, device=dev)).to(torch.bfloat16)  # [d, f] row-major = MN-major B for dY.W2
kw = dict(M=M, N=f, K=d, a_layout=0, b_layout=1, lda=d, ldb=f, ldc=f)
print("fc2 data gradient [3992 x 7680 x 1920], NN")
print(f"  plain                         {timeit(lambda: ops.gemm(dh, W2, g, **kw)):7.1f} us")
print(f"  + residual read               {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_RESIDUAL, **kw)):7.1f} us")
print(f"  + GELU'(R)                    {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_DGELU, **kw)):7.1f} us")
print(f"  + GELU'(R) + dropout          {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_DGELU, dropout_p=0.1, dropout_seed=7, **kw)):7.1f} us")
