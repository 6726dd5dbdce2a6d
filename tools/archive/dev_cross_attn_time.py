"""Decode cross-attention kernels alone (whisper-medium shape), graph-replayed chains of 48 launches:
fused LayerNorm + q-projection + attention, the attention alone, with and without the key split.
    python tools/dev_cross_attn_time.py [B ...]"""
import os
import sys
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
d, H, Tk = 1024, 16, 1500
hd = d // H


def timed(fn, n=48, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for B in [int(a) for a in sys.argv[1:]] or [4, 8, 16, 32]:
    gen = torch.Generator(device=DEV).manual_seed(B)
    x = torch.randn(B, d, device=DEV, generator=gen).to(torch.bfloat16)
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    W = (0.05 * torch.randn(d, d, device=DEV, generator=gen)).to(torch.bfloat16)
    bias = torch.zeros(d, device=DEV)
    # 24 layers' worth of K|V so that the chain does not re-read a cached 49 MB
    kvs = [torch.randn(B, Tk, 2 * d, device=DEV, generator=gen).to(torch.bfloat16) for _ in range(6)]
    q = torch.randn(B, d, device=DEV, generator=gen).to(torch.bfloat16)
    out = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros(B * H * 32, device=DEV)
    ws = ops.attn_split_workspace(B, H, DEV)
    akw = dict(B=B, H=H, Tk=Tk, hd=hd, scale=hd ** -0.5, ldk=2 * d, ldv=2 * d, ldo=d, skb=Tk * 2 * d, svb=Tk * 2 * d, sob=d, k_off=0, v_off=d)
    i = [0]

    def kv():
        i[0] = (i[0] + 1) % len(kvs)
        return kvs[i[0]]

    res = {}
    for name, split in (("one", None), ("split", ws)):
        def fused():
            k = kv()
            ops.decode_attn_qproj(x, gamma, beta, W, bias, k, k, out, d_model=d, eps=1e-5, ldx=d, ldw=d, split_ws=split, **akw)

        def plain():
            k = kv()
            ops.attn_fwd(q, k, k, out, lse, Tq=1, Tqp=32, ldq=d, sqb=d, split_ws=split, **akw)

        res[name] = (timed(fused), timed(plain))
    mb = B * Tk * 2 * d * 2 / 1e6
    print(f"B={B:3d} ({mb:5.1f} MB of K|V)  fused: one {res['one'][0]:6.2f} us, split {res['split'][0]:6.2f} | attention alone: one {res['one'][1]:6.2f}, "
          f"split {res['split'][1]:6.2f}  (K|V at 5 TB/s: {mb / 5:5.1f} us)")
