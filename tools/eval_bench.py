#!/usr/bin/env python3
"""Dev tool: time the evaluation kernels on a random [512, 21841] logits matrix."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from hgr_net_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, N, L = 512, 21841, 12
ld = (N + 63) // 64 * 64
lg = (torch.randn(B, ld, device="cuda") * 0.05)[:, :N]
depth = torch.randint(0, L, (N,), dtype=torch.int32, device="cuda")
perm = torch.randperm(N)
train = torch.arange(N, dtype=torch.int32, device="cuda")
test = perm[:13442].to(torch.int32).cuda()
idx = ops.EvalIndex(depth, train, test, L)
idx0 = ops.EvalIndex(depth, train, None, L)
print("eval_rows k=20 :", round(timeit(lambda: ops.eval_rows(lg, idx, 20)), 1), "us")
print("eval_rows k=1  :", round(timeit(lambda: ops.eval_rows(lg, idx, 1)), 1), "us")
print("eval_rows k=0  :", round(timeit(lambda: ops.eval_rows(lg, idx0, 0)), 1), "us")
print("level_argmax   :", round(timeit(lambda: ops.level_argmax(lg, depth, L, cols=train, want_top1=True)), 1), "us")
print("topk_rows k=20 :", round(timeit(lambda: ops.topk_rows(lg, 20, cols=test)), 1), "us")
print("bytes/row read once:", N * 4 * B / 1e6, "MB")
