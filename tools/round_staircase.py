#!/usr/bin/env python3
"""Dev tool: time of the residual-producer GEMM (hgr_gemm_nt_res_stats, N = 768, K = 768 / 3072) against the number of 256-row
panels: a staircase shows how much of a launch is the partly filled last round of workgroup slots (512 slots = 256 CUs x 2)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops

dt, w = torch.float16, 768
panels = [int(x) for x in sys.argv[1:]] or [40, 60, 80, 84, 85, 86, 90, 100, 110, 120, 128, 140, 160, 170, 171, 180, 200]


def timeit(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


torch.manual_seed(0)
mmax = max(panels) * 256
for name, k in (("out", w), ("proj", 4 * w)):
    a = (torch.randn(mmax, k, device="cuda") * 0.5).to(dt)
    wt = (torch.randn(w, k, device="cuda") * 0.03).to(dt)
    b = torch.randn(w, device="cuda")
    xh = torch.randn(mmax, w, device="cuda").to(dt)
    xl = torch.full((mmax, w), 128, device="cuda", dtype=torch.uint8)
    stats = torch.empty(mmax, w // 64, 2, device="cuda")
    res = {}
    for p in panels:
        m = p * 256
        f = lambda: ops.gemm_nt_res_stats(a[:m], wt, xh[:m], xl[:m], b, stats[:m])
        f(); f()
        t = min(timeit(f) for _ in range(5))
        res[p] = round(t, 1)
    print(json.dumps({"name": name, "tiles_per_panel": 6, "us_by_panels": res, "us_per_panel": {p: round(t / p, 3) for p, t in res.items()}}), flush=True)
