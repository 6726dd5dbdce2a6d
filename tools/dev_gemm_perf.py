"""Run one GEMM shape/variant repeatedly (for rocprofv3 --pmc passes and quick tuning).
usage: python tools/dev_gemm_perf.py M N K al bl [iters] [force]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

M, N, K, al, bl = (int(x) for x in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
force = int(sys.argv[7]) if len(sys.argv) > 7 else 0
f32 = int(sys.argv[8]) if len(sys.argv) > 8 else 0
pad = int(sys.argv[9]) if len(sys.argv) > 9 else 0  # extra elements in the leading dimensions (stride experiments)
dev = "cuda:0"
ops.lib().ca_gemm_force_kernel(force)
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = torch.randn(N, K, device=dev).to(torch.bfloat16)
def padded(t):
    r, c = t.shape
    buf = torch.zeros(r, c + pad, dtype=t.dtype, device=t.device)
    buf[:, :c] = t
    return buf


if al:
    A = A.t().contiguous()
if bl:
    B = B.t().contiguous()
if pad:
    A, B = padded(A), padded(B)
Cd = torch.zeros(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
kw = dict(M=M, N=N, K=K, a_layout=al, b_layout=bl, lda=(M if al else K) + pad, ldb=(N if bl else K) + pad, ldc=N,
          accumulate=bool(f32))
for _ in range(3):
    ops.gemm(A, B, Cd, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.gemm(A, B, Cd, **kw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"M{M} N{N} K{K} al{al} bl{bl} force{force} f32acc{f32} pad{pad}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.1f} TFLOP/s")
