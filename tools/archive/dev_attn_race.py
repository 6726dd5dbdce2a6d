"""Determinism stress of the fused-attention kernels: repeated launches must reproduce the first launch bit for bit."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
for (B, H, T, hd, pad) in ((2, 2, 499, 64, True), (2, 2, 499, 120, True), (8, 16, 499, 120, False), (2, 3, 130, 80, False),
                           (8, 16, 1500, 64, False)):
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device=dev).to(torch.bfloat16)
    dO = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
    Tqp = (T + 31) // 32 * 32
    klen = torch.tensor([T] + [max(1, T * 3 // 5)] * (B - 1), dtype=torch.int32, device=dev) if pad else None
    kw = dict(B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=Tqp, scale=hd ** -0.5, ldo=d, sob=T * d, klen=klen, causal=False,
              ldq=3 * d, ldk=3 * d, ldv=3 * d, sqb=T * 3 * d, skb=T * 3 * d, svb=T * 3 * d, q_off=0, k_off=d, v_off=2 * d)
    bkw = dict(lddo=d, sdob=T * d, lddq=3 * d, lddk=3 * d, lddv=3 * d, sdqb=T * 3 * d, sdkb=T * 3 * d, sdvb=T * 3 * d,
               dq_off=0, dk_off=d, dv_off=2 * d)
    ref = None
    bad_f = bad_b = 0
    n = 20
    for it in range(n):
        O = torch.zeros(B, T, d, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(B, H, Tqp, device=dev)
        Dq = torch.zeros(B, H, Tqp, device=dev)
        dqkv = torch.zeros_like(qkv)
        ops.attn_fwd(qkv, qkv, qkv, O, lse, **kw)
        ops.attn_bwd(qkv, qkv, qkv, O, lse, dO, Dq, dqkv, dqkv, dqkv, **bkw, **kw)
        if ref is None:
            ref = (O.clone(), dqkv.clone())
        else:
            if not torch.equal(O, ref[0]):
                bad_f += 1
                if bad_f <= 2:
                    df = (O.float() - ref[0].float()).abs()
                    idx = df.nonzero()
                    print("   fwd diff: n", idx.shape[0], "max", float(df.max()), "first", idx[0].tolist(), "last", idx[-1].tolist(), "lse nan", bool(torch.isnan(lse).any()))
            bad_b += int(not torch.equal(dqkv, ref[1]))
            if it == 1:
                ref = (O.clone(), dqkv.clone())
                bad_f = bad_b = 0
    print(f"B{B} H{H} T{T} hd{hd} pad{pad}: fwd mismatches {bad_f}/{n - 1}  bwd mismatches {bad_b}/{n - 1}", flush=True)
