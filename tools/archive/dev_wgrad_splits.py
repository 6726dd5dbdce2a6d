"""Conv-stack weight gradients (dW[512, 1536] = dY^T X over K = clips x frames rows): time of the split-K batched GEMM + the
fixed-order reduction for several split counts.   python tools/dev_wgrad_splits.py"""
import os
import sys
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402
from coral_amd.ops import MNMAJOR  # noqa: E402

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


for M, N, K in ((512, 1536, 204736), (512, 1536, 102336), (512, 1024, 31936), (512, 1536, 25592), (512, 1536, 12792), (1024, 512, 31936), (1024, 512, 3992)):
    dY = torch.randn(K, M, device=dev).to(torch.bfloat16)
    X = torch.randn(K, N, device=dev).to(torch.bfloat16)
    G = torch.zeros(M * N, dtype=torch.float32, device=dev)
    line = f"M={M} N={N} K={K}: rule -> {ops._wgrad_splits(M, N, K)} |"
    for s in (1, 4, 8, 16, 32):
        if K % s or K // s < 256:
            continue
        Kc = K // s
        ws = torch.empty(s * M * N, dtype=torch.float32, device=dev)

        def run():
            if s == 1:
                ops.gemm(dY, X, G, M=M, N=N, K=K, a_layout=MNMAJOR, lda=M, b_layout=MNMAJOR, ldb=N, ldc=N, out_f32=True)
            else:
                ops.gemm(dY, X, ws, M=M, N=N, K=Kc, a_layout=MNMAJOR, lda=M, b_layout=MNMAJOR, ldb=N, ldc=N, out_f32=True,
                         batch2=s, sA=(0, Kc * M), sB=(0, Kc * N), sC=(0, M * N))
                ops.reduce_rows(ws, s, M * N, M * N, G, accumulate=False)

        line += f" s={s}: {timeit(run):7.1f} us"
    print(line)
