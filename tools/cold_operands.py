#!/usr/bin/env python3
"""Dev tool: what a tower GEMM loses when its operands are not warm in L2 / Infinity Cache, as inside the step (every layer has its own
weights; the activations were written by the previous kernel and ~1 GB per layer streams through the 256 MB Infinity Cache).
Back-to-back launches of one shape with (a) the same W and A every time, (b) W rotating over 12 tensors, (c) W and A rotating
(A over 8 tensors = 315 MB for K = 768), (d) everything rotating + a 512 MB streaming write between launches timed separately."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_BIAS, EPI_BIAS_QUICKGELU

m = 25600
dt = torch.float16


def timeit(fn, iters=24):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, n, k, epi in (("qkv", 2304, 768, EPI_BIAS), ("fc", 3072, 768, EPI_BIAS_QUICKGELU), ("proj16", 768, 3072, EPI_BIAS)):
    torch.manual_seed(0)
    As = [(torch.rand(m, k, device="cuda") - 0.5).to(dt) for _ in range(8 if k == 768 else 4)]
    Ws = [((torch.rand(n, k, device="cuda") - 0.5) * 0.1).to(dt) for _ in range(12)]
    bias = torch.rand(n, device="cuda")
    out = torch.empty(m, n, dtype=dt, device="cuda")
    variants = {
        "same": lambda i: ops.gemm_nt(As[0], Ws[0], out, bias=bias, epilogue=epi),
        "w_rot": lambda i: ops.gemm_nt(As[0], Ws[i % 12], out, bias=bias, epilogue=epi),
        "aw_rot": lambda i: ops.gemm_nt(As[i % len(As)], Ws[i % 12], out, bias=bias, epilogue=epi),
    }
    res = {}
    for rep in range(3):
        for kname, f in variants.items():
            f(0); f(1)
            res.setdefault(kname, []).append(timeit(f))
    print(json.dumps({"shape": name, **{k_: round(min(v), 1) for k_, v in res.items()}}), flush=True)
