#!/usr/bin/env python3
"""Dev tool: weight-gradient product dW = dY^T X, old route (two transposes + NT split-K) vs hgr_gemm_tn_splitk."""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib
importlib.import_module("hgr_net_amd")
from hgr_net_amd import ops
def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
# (m, n, k): ViT-B/32 B=256 linears, text tower, RN 1x1 convs
SHAPES = [(12800, 2304, 768), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072), (40000, 1536, 512), (40000, 2048, 512),
          (802816, 64, 256), (802816, 256, 64), (200704, 512, 128), (50176, 1024, 256), (12544, 2048, 512), (12544, 512, 2048)]
for m, n, k in SHAPES:
    dy = torch.randn(m, n, device="cuda").bfloat16(); x = torch.randn(m, k, device="cuda").bfloat16()
    dyt = torch.empty(n, m, device="cuda", dtype=torch.bfloat16); xt = torch.empty(k, m, device="cuda", dtype=torch.bfloat16)
    row = {"shape": [m, n, k]}
    for tiles_target in (256, 512, 1024):
        tiles = -(-n // 128) * -(-k // 128)
        s = max(1, min(m // 64, -(-tiles_target // tiles)))
        kc = (-(-m // s) + 63) // 64 * 64
        s = -(-m // kc)
        part = torch.empty(s, n * k, device="cuda", dtype=torch.float32)
        us = min(timeit(lambda: ops.gemm_tn_splitk(dy, x, part, kc)) for _ in range(3))
        row[f"tn{tiles_target}_us"] = round(us, 1); row[f"tn{tiles_target}_s"] = s
    # old route
    big = n >= 256 and k >= 256
    tiles = (-(-n // 256) * -(-k // 256)) if big else (-(-n // 128) * -(-k // 128))
    s = max(1, min(m // (128 if big else 64), -(-(256 if big else 512) // tiles)))
    kc = (-(-m // s) + 63) // 64 * 64
    s = -(-m // kc)
    part = torch.empty(s, n * k, device="cuda", dtype=torch.float32)
    t_tr = min(timeit(lambda: (ops.transpose16(dy, dyt), ops.transpose16(x, xt))) for _ in range(3))
    t_nt = min(timeit(lambda: ops.gemm_nt_splitk(dyt, xt, part, kc)) for _ in range(3))
    row["old_transposes_us"] = round(t_tr, 1); row["old_nt_us"] = round(t_nt, 1); row["old_s"] = s
    row["tflops_tn_best"] = round(2.0 * m * n * k / min(row["tn256_us"], row["tn512_us"], row["tn1024_us"]) / 1e6)
    print(json.dumps(row))
