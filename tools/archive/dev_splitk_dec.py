"""Whisper decoder GEMMs (M = 896 teacher-forced rows): plain launch against split-K as a batched launch into fp32 partials
(the reduction not included):  python tools/dev_splitk_dec.py"""
import os
import sys
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402
from coral_amd.ops import MNMAJOR  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 896
for N, K, mn in ((1024, 4096, False), (1024, 4096, True), (1024, 1024, False), (3072, 1024, False), (1280, 5120, False)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(K, N, device=dev) if mn else torch.randn(N, K, device=dev)).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(b_layout=MNMAJOR, ldb=N) if mn else dict(ldb=K)
    t0 = timeit(lambda: ops.gemm(A, W, out, M=M, N=N, K=K, lda=K, ldc=N, **kw))
    line = f"M={M} N={N} K={K} {'B MN-major (data gradient)' if mn else 'B K-major (forward)'}: plain {t0:6.1f} us"
    for S in (2, 4):
        Kc = K // S
        part = torch.zeros(S, M, N, dtype=torch.float32, device=dev)
        sB = (Kc * N, 0) if mn else (Kc, 0)
        t = timeit(lambda: ops.gemm(A, W, part, M=M, N=N, K=Kc, lda=K, ldc=N, batch1=S, sA=(Kc, 0), sB=sB, sC=(M * N, 0),
                                    out_f32=True, **kw))
        ref = (A.float() @ (W.float() if mn else W.float().t()))
        err = float((part.sum(0) - ref).abs().max() / ref.abs().max())
        line += f" | split {S}: {t:6.1f} us (err {err:.1e})"
    print(line)
