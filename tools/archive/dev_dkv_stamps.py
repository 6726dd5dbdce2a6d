"""Where the dK|dV kernel's cycles go per 32-query step (needs the -DDKV_STAMPS build of attention.hip:
CORAL_AMD_LIB=coral_amd/libvariant_dkvst.so).  usage: python tools/dev_dkv_stamps.py"""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops  # noqa: E402

dev = "cuda:0"
ops.lib()
raw = ctypes.CDLL(str(Path(os.environ["CORAL_AMD_LIB"]).resolve()))
for name, B, H, T, hd in (("whisper-turbo enc", 8, 20, 1500, 64), ("xlsr-300m", 8, 16, 499, 64)):
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device=dev).to(torch.bfloat16)
    dqkv = torch.zeros_like(qkv)
    O = torch.zeros(B, T, d, dtype=torch.bfloat16, device=dev)
    dO = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
    Tqp = (T + 31) // 32 * 32
    lse = torch.zeros(B, H, Tqp, device=dev)
    Dq = torch.zeros(B, H, Tqp, device=dev)
    kw = dict(B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=Tqp, scale=hd ** -0.5, ldo=d, sob=T * d, klen=None, causal=False,
              ldq=3 * d, ldk=3 * d, ldv=3 * d, sqb=T * 3 * d, skb=T * 3 * d, svb=T * 3 * d, q_off=0, k_off=d, v_off=2 * d)
    bkw = dict(lddo=d, sdob=T * d, lddq=3 * d, lddk=3 * d, lddv=3 * d, sdqb=T * 3 * d, sdkb=T * 3 * d, sdvb=T * 3 * d,
               dq_off=0, dk_off=d, dv_off=2 * d)
    ops.attn_fwd(qkv, qkv, qkv, O, lse, **kw)
    for _ in range(3):
        ops.attn_bwd(qkv, qkv, qkv, O, lse, dO, Dq, dqkv, dqkv, dqkv, **bkw, **kw)
    torch.cuda.synchronize()
    n = 2048 * 4 * 8
    buf = (ctypes.c_longlong * n)()
    raw.ca_dkv_stamps(buf, n)
    s = np.frombuffer(buf, dtype=np.int64).reshape(2048, 4, 8).astype(np.float64)
    nwg = min(2048, ((T + 127) // 128) * H * B)
    s = s[:nwg]
    steps = s[:, :, 5]
    ok = steps > 0
    per = [(s[:, :, i][ok] / steps[ok]).mean() for i in range(5)]
    print(f"{name}: cycles per 32-query step and wave: wait+barrier {per[0]:.0f} | issue next {per[1]:.0f} | S, dP MFMAs (16) {per[2]:.0f} | "
          f"softmax / dS VALU + first transposed reads {per[3]:.0f} | dV, dK MFMAs (16) {per[4]:.0f} | total {sum(per):.0f}  (MFMA floor 512)")
