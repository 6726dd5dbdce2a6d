"""fp8 forward GEMM against the bf16 kernels on the Whisper-large-turbo / XLS-R forward shapes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops
dev = "cuda:0"


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in ((12000, 1280, 1280), (12000, 3840, 1280), (12000, 5120, 1280), (12000, 1280, 5120), (3992, 7680, 1920), (3992, 1920, 7680), (8192, 8192, 8192)):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    xq = torch.zeros(M, K, dtype=torch.uint8, device=dev); wq = torch.zeros(N, K, dtype=torch.uint8, device=dev)
    sx = torch.zeros(1, device=dev); sw = torch.zeros(1, device=dev); ws = torch.zeros(1, device=dev)
    tq = t(lambda: ops.quantize_fp8(x, xq, sx, ws))
    ops.quantize_fp8(w, wq, sw, ws)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    tb = t(lambda: ops.gemm(x, w, out, **kw))
    t8 = t(lambda: ops.gemm_fp8(xq, wq, out, a_scale=sx, b_scale=sw, **kw))
    fl = 2 * M * N * K / 1e6
    print(f"M{M} N{N} K{K}: bf16 {tb:7.1f} us {fl/tb:7.1f} TF | fp8 {t8:7.1f} us {fl/t8:7.1f} TF | quantize A {tq:6.1f} us")
