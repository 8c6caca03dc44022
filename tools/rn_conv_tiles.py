#!/usr/bin/env python3
"""Dev tool: RN50's 3x3 convolutions at batch 512 under each forced tile (0 = host plan)."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib
importlib.import_module("hgr_net_amd")
from hgr_net_amd import ops
def timeit(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 512
for h, c, cout in [(56, 128, 128), (28, 128, 128), (28, 256, 256), (14, 256, 256), (14, 512, 512), (7, 512, 512), (56, 64, 64), (112, 32, 32), (112, 32, 64)]:
    x = torch.randn(B * h * h, c, device="cuda").half(); k = 9 * c; kp = (k + 63) // 64 * 64
    w = (torch.randn(cout, kp, device="cuda") * 0.05).half(); bias = torch.randn(cout, device="cuda")
    out = torch.empty(B * h * h, cout, device="cuda", dtype=torch.float16)
    row = {"hw": h, "c": c, "cout": cout}
    for tile in (0, 128, 256):
        ops.gemm_set_tile(tile)
        us = min(timeit(lambda: ops.conv3x3_nhwc(x, w, bias, out, B, h, h, c)) for _ in range(3))
        row[f"t{tile}_us"] = round(us, 1); row[f"t{tile}_tf"] = round(2.0 * B * h * h * cout * k / us / 1e6)
    ops.gemm_set_tile(0)
    print(json.dumps(row))
