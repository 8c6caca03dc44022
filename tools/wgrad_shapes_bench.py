#!/usr/bin/env python3
"""Dev tool: the weight-gradient GEMM (hgr_gemm_tn_splitk + the slice reduction) and the data-gradient GEMM of every Linear shape of the
ViT-L/14 training step at batch 256 (vision M = 65 792, text M = 81 397), back to back: time, TF/s, slice plan."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

from hgr_net_amd import ops
DEV="cuda"
def t(fn,n=10):
    for i in range(2): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
dt=torch.bfloat16
for (m,n,k,name) in ((65792,3072,1024,"v qkv"),(65792,1024,1024,"v out"),(65792,4096,1024,"v fc"),(65792,1024,4096,"v proj"),
                     (81397,2304,768,"t qkv"),(81397,768,768,"t out"),(81397,3072,768,"t fc"),(81397,768,3072,"t proj")):
    dy=torch.randn(m,n,device=DEV).to(dt); x=torch.randn(m,k,device=DEV).to(dt)
    s=ops.tn_slices(n,k,m); kc=(-(-m//s)+63)//64*64; s=-(-m//kc)
    part=torch.empty(s,n*k,dtype=torch.float32,device=DEV)
    us=t(lambda: ops.gemm_tn_splitk(dy,x,part,kc))
    gw=torch.zeros(n*k,device=DEV); scr=torch.empty(max(1<<22,n*k),device=DEV)
    us2=t(lambda: ops.colsum(part,gw,scr,accumulate=True))
    print(f"{name}: dW [{n}x{k}] over M={m}: slices {s} kc {kc}: gemm_tn {us:.1f} us = {2*m*n*k/us/1e6:.0f} TF/s; reduce {us2:.1f} us", flush=True)
    # dX for comparison: dy [m,n] @ w[n,k] -> [m,k] : gemm_nt(dy, wt[k,n])
    wt=torch.randn(k,n,device=DEV).to(dt); dx=torch.empty(m,k,dtype=dt,device=DEV)
    us3=t(lambda: ops.gemm_nt(dy,wt,dx))
    print(f"      dX gemm_nt {us3:.1f} us = {2*m*n*k/us3/1e6:.0f} TF/s", flush=True)
