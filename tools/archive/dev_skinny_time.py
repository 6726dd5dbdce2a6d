"""Latency of the skinny GEMM (one decoded token per clip): hot (same weight every launch) vs cold (a different weight
matrix per launch, 1.6 GB cycled), eager launches vs one HIP-graph replay."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
M = 8


def run(N, K, nmat, ln, graph, iters=192):
    x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(nmat)]
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    bias = torch.zeros(N, device=dev)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias)

    def body():
        for i in range(iters):
            ops.gemm(x, Ws[i % nmat], out, **kw)

    body()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                body()
        fn = g.replay
    else:
        fn = body
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * iters) * 1e3


for N, K in ((1024, 1024), (4096, 1024), (1024, 4096)):
    nbytes = N * K * 2
    ncold = max(2, int(1.6e9 // nbytes))
    for ln in (False,):
        r = [run(N, K, nm, ln, gr) for nm in (1, min(ncold, 192)) for gr in (False, True)]
        print(f"N{N} K{K} ln={int(ln)}: hot eager {r[0]:.1f} us, hot graph {r[1]:.1f} us, cold eager {r[2]:.1f} us, cold graph {r[3]:.1f} us")
