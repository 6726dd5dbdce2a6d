"""What the fused FFN epilogues cost on top of the plain GEMM of the same shape (XLS-R-2B, kernel X):  python tools/dev_epilogue_cost.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"


class Pair(float):
    """Time with the specialised epilogue; formats as 'fast us (general walk: x us)'."""

    def __new__(cls, fast, general):
        o = float.__new__(cls, fast)
        o.general = general
        return o

    def __format__(self, spec):
        return f"{float(self):{spec}} us   (general walk {self.general:{spec}})"


def timeit(fn, iters=20):
    out = []
    for general in (0, 1):
        ops.lib().ca_gemm_debug_general_epilogue(general)
        out.append(_timeit(fn, iters))
    ops.lib().ca_gemm_debug_general_epilogue(0)
    return Pair(out[0], out[1])


def _timeit(fn, iters=20):
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
print(f"  plain                         {timeit(lambda: ops.gemm(x, W1, u, **kw)):7.1f}")
print(f"  + bias                        {timeit(lambda: ops.gemm(x, W1, u, bias=b1, **kw)):7.1f}")
print(f"  + bias + residual read        {timeit(lambda: ops.gemm(x, W1, u, bias=b1, R=R, ldr=f, epilogue=ops.EPI_RESIDUAL, **kw)):7.1f}")
print(f"  + bias + GELU, 2 outputs      {timeit(lambda: ops.gemm(x, W1, u, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, **kw)):7.1f}")
print(f"  + bias + GELU + dropout, 2 out{timeit(lambda: ops.gemm(x, W1, u, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, dropout_p=0.1, dropout_seed=7, **kw)):7.1f}")
print(f"  + bias + GELU + dropout, g only{timeit(lambda: ops.gemm(x, W1, None, bias=b1, C2=g, c2_off=0, epilogue=ops.EPI_GELU, dropout_p=0.1, dropout_seed=7, **kw)):6.1f}")
dh = torch.randn(M, d, device=dev).to(torch.bfloat16)
W2 = (0.02 * torch.randn(d, f, device=dev)).to(torch.bfloat16)  # [d, f] row-major = MN-major B for dY.W2
kw = dict(M=M, N=f, K=d, a_layout=0, b_layout=1, lda=d, ldb=f, ldc=f)
print("fc2 data gradient [3992 x 7680 x 1920], NN")
print(f"  plain                         {timeit(lambda: ops.gemm(dh, W2, g, **kw)):7.1f}")
print(f"  + residual read               {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_RESIDUAL, **kw)):7.1f}")
print(f"  + GELU'(R)                    {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_DGELU, **kw)):7.1f}")
print(f"  + GELU'(R) + dropout          {timeit(lambda: ops.gemm(dh, W2, g, R=R, ldr=f, epilogue=ops.EPI_DGELU, dropout_p=0.1, dropout_seed=7, **kw)):7.1f}")
# the N = d shapes on kernel L (out-projection, fc2 forward with residual; q|k|v with bias) and the weight gradient
W2f = (0.02 * torch.randn(d, f, device=dev)).to(torch.bfloat16)
gact = torch.randn(M, f, device=dev).to(torch.bfloat16)
h1 = torch.randn(M, d, device=dev).to(torch.bfloat16)
hout = torch.empty(M, d, dtype=torch.bfloat16, device=dev)
b2 = torch.zeros(d, device=dev)
kw = dict(M=M, N=d, K=f, a_layout=0, b_layout=0, lda=f, ldb=f, ldc=d)
print("fc2 forward [3992 x 1920 x 7680], NT (kernel L)")
print(f"  plain                         {timeit(lambda: ops.gemm(gact, W2f, hout, **kw)):7.1f}")
print(f"  + bias + residual             {timeit(lambda: ops.gemm(gact, W2f, hout, bias=b2, R=h1, ldr=d, epilogue=ops.EPI_RESIDUAL, **kw)):7.1f}")
Wqkv = (0.02 * torch.randn(3 * d, d, device=dev)).to(torch.bfloat16)
qkv = torch.empty(M, 3 * d, dtype=torch.bfloat16, device=dev)
b3 = torch.zeros(3 * d, device=dev)
kw = dict(M=M, N=3 * d, K=d, a_layout=0, b_layout=0, lda=d, ldb=d, ldc=3 * d)
print("q|k|v forward [3992 x 5760 x 1920], NT (kernel L)")
print(f"  + bias                        {timeit(lambda: ops.gemm(x, Wqkv, qkv, bias=b3, **kw)):7.1f}")
G = torch.zeros(f * d, dtype=torch.float32, device=dev)
slots = torch.zeros(ops.sumsq_slots(f, d), dtype=torch.float32, device=dev)
du = torch.randn(M, f, device=dev).to(torch.bfloat16)
print("fc1 weight gradient [7680 x 1920 x 3992], TN, fp32 out")
print(f"  written + sums of squares     {timeit(lambda: ops.wgrad_gemm(du, x, G, M=f, N=d, K=M, lda=f, ldb=d, c_off=0, accumulate=False, sq=(slots, 0))):7.1f}")
print(f"  accumulated                   {timeit(lambda: ops.wgrad_gemm(du, x, G, M=f, N=d, K=M, lda=f, ldb=d, c_off=0, accumulate=True)):7.1f}")
