"""How far ahead of the GPU can the host run?  Enqueue N launches of a ~300 us GEMM and time the host loop."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops
dev = "cuda:0"
M, N, K = 3992, 7680, 1920
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = torch.randn(N, K, device=dev).to(torch.bfloat16)
C = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
for _ in range(3):
    ops.gemm(A, B, C, **kw)
torch.cuda.synchronize()
for n in (8, 32, 128, 512, 1024, 2048):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ops.gemm(A, B, C, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n:5d}: host loop {(t1 - t0) * 1e3:8.2f} ms ({(t1 - t0) / n * 1e6:6.1f} us/launch), GPU done after {(t2 - t0) * 1e3:8.2f} ms")
x = torch.zeros(8, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    x = x + 1
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"torch tiny op: {(t1 - t0) / 200 * 1e6:.1f} us/op host")
