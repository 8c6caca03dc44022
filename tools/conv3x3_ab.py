#!/usr/bin/env python3
"""Dev tool: the implicit-GEMM 3x3 convolutions of the ModifiedResNet RN50 tower at batch 512 (bias + ReLU, stride 1) of the library
in HGR_LIB (default: the tree's); alternate builds with tools/lib_ab.sh.  Back-to-back launches, min / median of 5 x 6, us + TFLOP/s."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops

DEV, B = "cuda", 512
g = torch.Generator(device=DEV).manual_seed(4)
res = {}


def timeit(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(iters):
        fn(k)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


warm = False
for name, h, c in (("l1_56_c64", 56, 64), ("l2_56_c128", 56, 128), ("l2_28_c128", 28, 128), ("l3_28_c256", 28, 256), ("l3_14_c256", 14, 256),
                   ("l4_14_c512", 14, 512), ("l4_7_c512", 7, 512)):
    x = [torch.randn((B * h * h, c), generator=g, device=DEV).half() for _ in range(2)]
    w = ((2.0 / (9 * c)) ** 0.5 * torch.randn((c, 9 * c), generator=g, device=DEV)).half()
    bias = torch.randn((c,), generator=g, device=DEV)
    out = torch.empty((B * h * h, c), dtype=torch.float16, device=DEV)
    fn = lambda k: ops.conv3x3_nhwc(x[k & 1], w, bias, out, B, h, h, c)
    if not warm:
        for _ in range(20):
            timeit(fn)
        warm = True
    fn(0)
    ts = sorted(timeit(fn) for _ in range(5))
    res[name] = [round(ts[0], 1), round(ts[2], 1), round(2.0 * B * h * h * c * 9 * c / ts[0] / 1e6), float(out.float().abs().sum())]
    del x, out
print(json.dumps({"lib": os.path.basename(os.environ.get("HGR_LIB", "tree")), "us_min_med_tflops_checksum": res}))
