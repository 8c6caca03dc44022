#!/usr/bin/env python3
"""Dev tool: the HBM-bound 1x1-convolution GEMMs of RN50 at batch 512 under each forced tile (0 = cost model)."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib
importlib.import_module("hgr_net_amd")
from hgr_net_amd import ops
SHAPES = [(1605632, 256, 64, 5), (401408, 512, 128, 5), (100352, 1024, 256, 5), (1605632, 64, 256, 4), (1605632, 256, 64, 1),
          (6422528, 32, 64, 4), (1605632, 128, 256, 4), (401408, 128, 512, 4), (25088, 2048, 512, 5),
          (401408, 512, 256, 1), (100352, 1024, 512, 1), (25088, 2048, 1024, 1), (401408, 256, 512, 4), (100352, 512, 1024, 4), (100352, 256, 1024, 4),
          (25088, 512, 2048, 4), (25600, 2048, 2048, 1)]
def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for m, n, k, epi in SHAPES:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * 0.05).half()
    bias = torch.randn(n, device="cuda"); res = torch.randn(m, n, device="cuda").half() if epi == 5 else None
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    byts = 2 * m * k + 2 * n * k + 2 * m * n * (2 if epi == 5 else 1)
    row = {"shape": [m, n, k, epi], "MB": round(byts / 1e6)}
    for tile in (0, 128, 256):
        ops.gemm_set_tile(tile)
        us = min(timeit(lambda: ops.gemm_nt(a, w, out, bias=bias, residual=res, epilogue=epi)) for _ in range(3))
        row[f"t{tile}_us"] = round(us, 1); row[f"t{tile}_TBs"] = round(byts / us / 1e6, 2)
    ops.gemm_set_tile(0)
    print(json.dumps(row))
