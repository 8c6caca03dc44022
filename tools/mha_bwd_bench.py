#!/usr/bin/env python3
"""Dev tool: hgr_mha_bwd (and hgr_mha forward) at the ViT tower shapes of the training bench, back to back.
    mha_bwd_bench.py [batch]     (ViT-L/14: L 257, 16 heads; ViT-B/16: L 197, 12 heads; ViT-B/32: L 50, 12 heads)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = torch.bfloat16


def timeit(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, L, heads in (("L14", 257, 16), ("B16", 197, 12), ("B32", 50, 12)):
    w = heads * 64
    torch.manual_seed(0)
    qkv = (torch.randn(batch * L, 3 * w, device="cuda") * 0.7).to(dt)
    do = (torch.randn(batch * L, w, device="cuda") * 0.5).to(dt)
    out = torch.empty(batch * L, w, dtype=dt, device="cuda")
    dqkv = torch.empty_like(qkv)
    ops.mha(qkv, out, batch, L, heads, False)
    f = lambda: ops.mha(qkv, out, batch, L, heads, False)
    g = lambda: ops.mha_bwd(qkv, out, do, dqkv, batch, L, heads, False)
    f(); g()
    tf, tb = min(timeit(f) for _ in range(3)), min(timeit(g) for _ in range(3))
    fl = 4.0 * L * L * 64 * heads * batch
    print(json.dumps({"shape": name, "batch": batch, "L": L, "heads": heads, "fwd_us": round(tf, 1), "fwd_tflops": round(fl / tf / 1e6, 1),
                      "bwd_us": round(tb, 1), "bwd_tflops_2.5x": round(2.5 * fl / tb / 1e6, 1)}), flush=True)

# text-tower shape of the training bench: ~3 552 distinct prompts x 23 tokens (L <= 32: the one-wave-per-head kernel), causal
for name, n, L, heads in (("txt B32", 3552, 23, 8), ("txt L14", 3552, 23, 12)):
    w = heads * 64
    qkv = (torch.randn(n * L, 3 * w, device="cuda") * 0.7).to(dt)
    do = (torch.randn(n * L, w, device="cuda") * 0.5).to(dt)
    out = torch.empty(n * L, w, dtype=dt, device="cuda")
    dqkv = torch.empty_like(qkv)
    ops.mha(qkv, out, n, L, heads, True)
    g = lambda: ops.mha_bwd(qkv, out, do, dqkv, n, L, heads, True)
    g()
    tb = min(timeit(g) for _ in range(3))
    print(json.dumps({"shape": name, "prompts": n, "L": L, "heads": heads, "bwd_us": round(tb, 1)}), flush=True)
