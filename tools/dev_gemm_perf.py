"""Run one GEMM shape/variant repeatedly (for rocprofv3 --pmc passes and quick tuning).
usage: python tools/dev_gemm_perf.py M N K al bl [iters] [force] [f32acc] [pad] [epi]
epi: 0 plain, 1 bias+GELU+dropout with both outputs (FFN1 forward), 3 bias-free DGELU with dropout (FFN2 dgrad),
2 bias+residual (FFN2 forward)"""
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
epi = int(sys.argv[10]) if len(sys.argv) > 10 else 0
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
if epi:
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    C2 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 1:
        kw.update(bias=bias, epilogue=ops.EPI_GELU, C2=C2, dropout_p=0.1, dropout_seed=7)
    elif epi == 2:
        kw.update(bias=bias, epilogue=ops.EPI_RESIDUAL, R=R, ldr=N)
    elif epi == 3:
        kw.update(epilogue=ops.EPI_DGELU, R=R, ldr=N, dropout_p=0.1, dropout_seed=7)
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
print(f"M{M} N{N} K{K} al{al} bl{bl} force{force} f32acc{f32} pad{pad} epi{epi}: {ms*1e3:.1f} us {2*M*N*K/ms/1e9:.1f} TFLOP/s")
