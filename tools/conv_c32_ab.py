#!/usr/bin/env python3
"""Dev tool: the direct 3x3 convolution for 32 input channels (ModifiedResNet stem conv2 / conv3 + pool at 112 x 112, batch 512) of the
library in HGR_LIB (default: the tree's); alternate builds with tools/lib_ab.sh.  Back-to-back launches, min / median of 5 x 8, us."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops

DEV, B, H = "cuda", 512, 112
g = torch.Generator(device=DEV).manual_seed(2)
x = [torch.randn((B * H * H, 32), generator=g, device=DEV).half() for _ in range(2)]
res = {}


def timeit(fn, iters=8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(iters):
        fn(k)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, cout, pool in (("conv2_32to32", 32, False), ("conv3_32to64_pool", 64, True)):
    w = (0.06 * torch.randn((cout, 320), generator=g, device=DEV)).half()
    w[:, 288:] = 0
    bias = torch.randn((cout,), generator=g, device=DEV)
    out = torch.empty((B * H * H // (4 if pool else 1), cout), dtype=torch.float16, device=DEV)
    if pool:
        fn = lambda k: ops.conv3x3_pool2_nhwc(x[k & 1], w, bias, out, B, H, H, 32)
    else:
        fn = lambda k: ops.conv3x3_nhwc(x[k & 1], w, bias, out, B, H, H, 32)
    for _ in range(10):
        timeit(fn)
    ts = sorted(timeit(fn) for _ in range(5))
    res[name] = [round(ts[0], 1), round(ts[2], 1), float(out.float().abs().sum())]
print(json.dumps({"lib": os.path.basename(os.environ.get("HGR_LIB", "tree")), "us_min_med_checksum": res}))
