#!/usr/bin/env python3
"""Dev tool: hgr_mha (+ statistics) and hgr_mha_bwd back to back at the three attention shapes of the training step (ViT-L/14 257 tokens,
ViT-B/32 50 tokens, trimmed prompts 23 tokens causal), with a hash of the outputs: two builds (HGR_LIB=...) can be compared for time
AND bits."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

from hgr_net_amd import ops
DEV="cuda"
def t(fn,n=20):
    for i in range(3): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for dt in (torch.bfloat16, torch.float16):
  for (b,l,heads,causal) in ((256,257,16,False),(512,50,12,False),(3539,23,12,True)):
    w=heads*64; m=b*l
    g=torch.Generator(device=DEV).manual_seed(1)
    qkv=torch.randn(m,3*w,device=DEV,generator=g).to(dt)
    out=torch.empty(m,w,dtype=dt,device=DEV)
    st=torch.empty(b,heads,l,2,dtype=torch.float32,device=DEV) if l>32 else None
    f=t(lambda: ops.mha(qkv,out,b,l,heads,causal,stats=st))
    dout=torch.randn(m,w,device=DEV,generator=g).to(dt); dqkv=torch.empty_like(qkv)
    bw=t(lambda: ops.mha_bwd(qkv,out,dout,dqkv,b,l,heads,causal,stats=st))
    import hashlib
    h=hashlib.sha256(out.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    h2=hashlib.sha256(dqkv.cpu().view(torch.int16).numpy().tobytes()).hexdigest()[:12]
    print(f"lib={os.path.basename(os.environ.get('HGR_LIB','default'))} {dt} b={b} l={l}: fwd {f:.1f} us  bwd {bw:.1f} us  out {h} dqkv {h2}", flush=True)
