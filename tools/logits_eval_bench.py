#!/usr/bin/env python3
"""Dev tool: hgr_logits_eval (both stages) against hgr_gemm_nt + hgr_eval_rows at the bench shape (512 x 21 841 x 512), graph
replays of back-to-back launches; HGR_LE_DBG=1..4 cuts the row stage short (timing of its parts).
(round 6: these switches exist in the lab build only - `make -C hgr-net_amd/csrc lab`, then HGR_LIB=$PWD/hgr-net_amd/lib/libhgr_lab.so)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from hgr_net_amd import ops, synth

rows, n, d, levels, k = 512, 21841, 512, 12, 20
torch.manual_seed(0)
f = torch.randn(rows, d); f = (f / f.norm(dim=1, keepdim=True)).half().cuda()
z = torch.randn(n, d); z = (z / z.norm(dim=1, keepdim=True)).half().cuda()
depth = synth.randint(3, "depth", n, 0, levels).astype(np.int32)
perm = np.argsort(synth.uniform(4, "perm", n), kind="stable").astype(np.int32)
train, test = perm[: n - n // 3].copy(), perm[n - n // 2:].copy()
index = ops.EvalIndex(torch.from_numpy(depth).cuda(), torch.from_numpy(train).cuda(), torch.from_numpy(test).cuda(), levels)
plan = ops.LogitsEvalPlan(index).bind(z)
lg = torch.empty(rows, (n + 63) // 64 * 64, dtype=torch.float32, device="cuda")


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(json.dumps({"fused_us": round(timeit(lambda: ops.logits_eval(f, plan, k)), 1),
                  "tile_us": round(timeit(lambda: ops.logits_eval(f, plan, k, stage="tile")), 1),
                  "row_us": round(timeit(lambda: ops.logits_eval(f, plan, k, stage="row")), 1),
                  "gemm_us": round(timeit(lambda: ops.gemm_nt(f, z, lg, n=n)), 1),
                  "gemm_plus_eval_rows_us": round(timeit(lambda: (ops.gemm_nt(f, z, lg, n=n), ops.eval_rows(lg[:, :n], index, k))), 1)}))
