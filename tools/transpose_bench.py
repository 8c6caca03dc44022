#!/usr/bin/env python3
"""Dev tool: correctness + bandwidth of hgr_transpose16 on the shapes the weight-gradient path uses."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops
for (r,c) in [(16448,1024),(16448,4096),(12800,768),(12800,3072),(802816,64),(3084,512),(300,200),(16448,1000)]:
    x=torch.randn(r,c,device='cuda').bfloat16(); ld=(r+63)//64*64
    y=torch.zeros(c,ld,dtype=torch.bfloat16,device='cuda')
    ops.transpose16(x,y); assert torch.equal(y[:,:r],x.t()) and not y[:,r:].any()
    torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.transpose16(x,y)
    e.record(); torch.cuda.synchronize(); us=s.elapsed_time(e)/20*1e3
    print(r,c,round(us,1),'us',round(2*r*c*2/us/1e6,2),'TB/s')
