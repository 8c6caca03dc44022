#!/usr/bin/env python3
"""Dev tool: the fused in_proj + attention kernel (hgr_gemm_nt_ln_mha) against hgr_gemm_nt_ln + hgr_mha, back to back over rotating
buffers, ViT-B/32 shape at batch 512.  With an experiment build (HGR_LIB=..., parts compiled out) this gave the phase ablations of
profiles/NOTES.md (GEMM main loop / LayerNorm epilogue / attention)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

from hgr_net_amd import ops
DEV="cuda"; dt=torch.float16
b,l,heads=512,50,12; w=heads*64; m=b*l
g=torch.Generator(device=DEV).manual_seed(1)
NB=6
xs=[torch.randn(m,w,device=DEV,generator=g).to(dt) for _ in range(NB)]
wf=[(torch.randn(3*w,w,device=DEV,generator=g)*w**-0.5).to(dt) for _ in range(NB)]
s=torch.randn(3*w,device=DEV); c=torch.randn(3*w,device=DEV)
stats=torch.zeros(m,w//64,2,device=DEV); stats[...,1]=64.0*1.0
x32=xs[0].float(); 
xlo=torch.empty(m,w,dtype=ops.PAIR_LO,device=DEV)
x16=torch.empty(m,w,dtype=dt,device=DEV)
ops.row_stats16(x32,x16,xlo,stats)
outs=[torch.empty(m,w,dtype=dt,device=DEV) for _ in range(NB)]
qkv=[torch.empty(m,3*w,dtype=dt,device=DEV) for _ in range(NB)]
def t(fn,n=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
fl=2*m*w*3*w
for rep in range(3):
    a=t(lambda i: ops.gemm_nt_ln_mha(xs[i%NB],wf[i%NB],outs[i%NB],s,c,stats,b,l,heads,False,1e-5))
    d=t(lambda i: ops.gemm_nt_ln(xs[i%NB],wf[i%NB],qkv[i%NB],s,c,stats,1e-5))
    e=t(lambda i: ops.mha(qkv[i%NB],outs[i%NB],b,l,heads,False))
    print(f"lib={os.environ.get('HGR_LIB','default')}: fused {a:.1f} us ({fl/a/1e6:.0f} TF/s)  duo qkv {d:.1f} us ({fl/d/1e6:.0f} TF/s)  mha {e:.1f} us", flush=True)
