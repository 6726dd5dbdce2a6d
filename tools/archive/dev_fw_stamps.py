import ctypes, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from coral_amd import ops
dev="cuda:0"; ops.lib()
raw=ctypes.CDLL(str(Path(os.environ["CORAL_AMD_LIB"]).resolve()))
for name,B,H,T,hd in (("whisper-turbo enc",8,20,1500,64),("xlsr-2b",8,16,499,120)):
    d=H*hd
    qkv=torch.randn(B,T,3*d,device=dev).to(torch.bfloat16); O=torch.zeros(B,T,d,dtype=torch.bfloat16,device=dev)
    Tqp=(T+31)//32*32; lse=torch.zeros(B,H,Tqp,device=dev)
    kw=dict(B=B,H=H,Tq=T,Tk=T,hd=hd,Tqp=Tqp,scale=hd**-0.5,ldo=d,sob=T*d,klen=None,causal=False,ldq=3*d,ldk=3*d,ldv=3*d,sqb=T*3*d,skb=T*3*d,svb=T*3*d,q_off=0,k_off=d,v_off=2*d)
    for _ in range(3): ops.attn_fwd(qkv,qkv,qkv,O,lse,**kw)
    torch.cuda.synchronize()
    n=2048*4*8; buf=(ctypes.c_longlong*n)(); raw.ca_fw_stamps(buf,n)
    s=np.frombuffer(buf,dtype=np.int64).reshape(2048,4,8).astype(np.float64)
    nwg=min(2048,((T+127)//128)*H*B); s=s[:nwg]; cnt=s[:,:,5]; ok=cnt>0
    per=[(s[:,:,i][ok]/cnt[ok]).mean() for i in range(5)]
    print(f"{name}: per 64-key tile and wave: wait+barrier {per[0]:.0f} | issue next {per[1]:.0f} | S MFMAs {per[2]:.0f} | softmax {per[3]:.0f} | PV MFMAs + V reads {per[4]:.0f} | total {sum(per):.0f}")
