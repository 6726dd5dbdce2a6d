"""Fused-attention kernel times at the models' shapes (tuning aid):  python tools/dev_attn_perf.py
CA_ATTN_WIDE_MIN=100000 selects the 64-query workgroups everywhere (A/B against the 128-query ones)."""
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


for name, B, H, T, hd in (("xlsr-2b", 8, 16, 499, 120), ("xlsr-1b", 8, 16, 499, 80), ("xlsr-300m", 8, 16, 499, 64),
                          ("whisper-medium enc", 8, 16, 1500, 64), ("whisper-turbo enc", 8, 20, 1500, 64)):
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device=dev).to(torch.bfloat16)
    dqkv = torch.zeros_like(qkv)
    O = torch.zeros(B, T, d, dtype=torch.bfloat16, device=dev)
    dO = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
    Tqp = (T + 31) // 32 * 32
    lse = torch.zeros(B, H, Tqp, device=dev)
    Dq = torch.zeros(B, H, Tqp, device=dev)
    klen = torch.full((B,), T, dtype=torch.int32, device=dev)
    kw = dict(B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=Tqp, scale=hd ** -0.5, ldo=d, sob=T * d, klen=klen, causal=False,
              ldq=3 * d, ldk=3 * d, ldv=3 * d, sqb=T * 3 * d, skb=T * 3 * d, svb=T * 3 * d, q_off=0, k_off=d, v_off=2 * d)
    bkw = dict(lddo=d, sdob=T * d, lddq=3 * d, lddk=3 * d, lddv=3 * d, sdqb=T * 3 * d, sdkb=T * 3 * d, sdvb=T * 3 * d,
               dq_off=0, dk_off=d, dv_off=2 * d)
    tf = timeit(lambda: ops.attn_fwd(qkv, qkv, qkv, O, lse, **kw))
    tb = timeit(lambda: ops.attn_bwd(qkv, qkv, qkv, O, lse, dO, Dq, dqkv, dqkv, dqkv, **bkw, **kw))
    fl = 4.0 * B * H * T * T * hd
    print(f"{name:20s} B{B} H{H} T{T} hd{hd}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF)   bwd (prep+dkv+dq) {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF)")
