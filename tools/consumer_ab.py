#!/usr/bin/env python3
"""Dev tool: what the LayerNorm-folded consumer epilogue costs over the plain bias epilogue of the same kernel (gemm_nt_duo), same
shapes (ViT-B/32 in_proj / c_fc at batch 512), one process, interleaved, rotating operands."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_BIAS, EPI_BIAS_QUICKGELU

DEV, M = "cuda", 25600
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda shape, scale=1.0: scale * torch.randn(shape, generator=g, device=DEV)
NB = 4
x = [rnd((M, 768)).half() for _ in range(NB)]
xl = torch.full((M, 768), 128, dtype=torch.uint8, device="cuda")     # the pair's low byte (round 4)
stats = torch.empty((M, 12, 2), dtype=torch.float32, device=DEV)
ops.row_stats16(rnd((M, 768)), x[0].clone(), xl, stats)
res = {}
for name, n, gelu in (("qkv", 2304, False), ("fc", 3072, True)):
    w = rnd((n, 768), 0.03).half()
    b, s_, c_ = rnd((n,)), rnd((n,)), rnd((n,))
    out = [torch.empty((M, n), dtype=torch.float16, device=DEV) for _ in range(2)]
    fns = {"bias": lambda i: ops.gemm_nt(x[i % NB], w, out[i & 1], bias=b, epilogue=EPI_BIAS_QUICKGELU if gelu else EPI_BIAS),
           "ln": lambda i: ops.gemm_nt_ln(x[i % NB], w, out[i & 1], s_, c_, stats, 1e-5, quickgelu=gelu)}
    ts = {k: [] for k in fns}
    for rep in range(7):
        for k, f in fns.items():
            f(0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(12):
                f(i)
            e.record()
            torch.cuda.synchronize()
            ts[k].append(s.elapsed_time(e) / 12 * 1e3)
    res[name] = {k: [round(min(v), 1), round(sorted(v)[len(v) // 2], 1)] for k, v in ts.items()}
print(json.dumps(res))
