#!/usr/bin/env python3
"""Dev tool: hgr_gemm_tn_splitk (weight gradients, operands as they lie in memory) at the ViT shapes of the training bench.
    HGR_TN_TILE=128|256 python3 wgrad_bench.py [rows]      (default rows = 256 x 257 = ViT-L/14 at batch 256)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import _lib, ops

m = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 257
dt = torch.bfloat16


def timeit(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, na, nb in (("L14 qkv", 3072, 1024), ("L14 out", 1024, 1024), ("L14 fc", 4096, 1024), ("L14 proj", 1024, 4096),
                     ("B32 qkv", 2304, 768), ("B32 out", 768, 768), ("B32 fc", 3072, 768), ("txt fc", 2048, 512)):
    torch.manual_seed(0)
    p = (torch.randn(m, na, device="cuda") * 0.1).to(dt)
    q = (torch.randn(m, nb, device="cuda") * 0.5).to(dt)
    s = ops.tn_slices(na, nb, m)
    kc = (-(-m // s) + 63) // 64 * 64
    s = -(-m // kc)
    part = torch.empty(s, na * nb, dtype=torch.float32, device="cuda")
    out = torch.zeros(na * nb, device="cuda")
    scratch = torch.empty(na * nb * 4, device="cuda")
    f = lambda: ops.gemm_tn_splitk(p, q, part.view(s, na, nb), kc)
    f(); f()
    t = min(timeit(f) for _ in range(3))
    print(json.dumps({"shape": name, "rows": m, "na": na, "nb": nb, "tile": int(_lib.load().hgr_gemm_tn_tile(na, nb)), "slices": s, "kc": kc,
                      "us": round(t, 1), "tflops": round(2.0 * m * na * nb / t / 1e6, 1)}), flush=True)
