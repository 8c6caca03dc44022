#!/usr/bin/env python3
"""Dev tool: the tile plans of hgr_gemm_nt against each other in ONE process, interleaved rounds (rule 24 of the CDNA
guide: box-to-box clock differences between gpurun calls are larger than the effects being measured).

    gemm_plan_ab.py [plans ...]        plans from {0 (cost model), 128, 256, 2 (256 x 128, two workgroups per CU)}

Per tower shape: bit-equality of every plan's output with plan 256's (same K order per output element, so the plans must
agree exactly), then min / median time over interleaved rounds of back-to-back launches; optionally hipBLASLt
(torch.matmul, bias / residual free) as a known-good reference on the same data."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL, EPI_NONE

plans = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [0, 256, 128, 2]
SHAPES = [("qkv", 25600, 2304, 768, EPI_BIAS, False), ("out", 25600, 768, 768, EPI_BIAS_RESIDUAL, True),
          ("fc", 25600, 3072, 768, EPI_BIAS_QUICKGELU, False), ("proj", 25600, 768, 3072, EPI_BIAS_RESIDUAL, True),
          ("patch", 25088, 768, 3072, EPI_NONE, True), ("logits", 512, 21841, 512, EPI_NONE, True),
          ("L14qkv", 131584, 3072, 1024, EPI_BIAS, False), ("L14out", 131584, 1024, 1024, EPI_BIAS_RESIDUAL, True),
          ("L14fc", 131584, 4096, 1024, EPI_BIAS_QUICKGELU, False), ("L14proj", 131584, 1024, 4096, EPI_BIAS_RESIDUAL, True),
          ("txtqkv", 8192 * 11, 1536, 512, EPI_BIAS, False), ("ragged", 1000, 777 + 3, 192, EPI_BIAS, False),
          ("sq4096", 4096, 4096, 4096, EPI_NONE, False),
          # ModifiedResNet 1x1 convolutions at batch 512 (5 = bias + 16-bit identity + ReLU, 4 = bias + ReLU)
          ("rn28_c3", 401408, 512, 128, 5, False), ("rn14_c3", 100352, 1024, 256, 5, False), ("rn7_c3", 25088, 2048, 512, 5, False),
          ("rn56_c1", 1605632, 64, 256, 4, False), ("rn28_c1", 401408, 128, 512, 4, False), ("rn14_c1", 100352, 256, 1024, 4, False)]
if "--rn" in sys.argv:
    SHAPES = [x for x in SHAPES if x[0].startswith("rn")]


def timeit(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, m, n, k, epi, o32 in SHAPES:
    torch.manual_seed(1)
    a = (torch.rand(m, k, device="cuda") * 2 - 1).half()
    w = ((torch.rand(n, k, device="cuda") * 2 - 1) * 0.05).half()
    bias = torch.rand(n, device="cuda") if epi != EPI_NONE else None
    ld = (n + 63) // 64 * 64
    res = torch.rand(m, ld, device="cuda") if epi == EPI_BIAS_RESIDUAL else (torch.rand(m, ld, device="cuda").half() if epi == 5 else None)
    outs = {}
    for pl in plans:                                  # correctness: separate residual, fresh output
        ops.gemm_set_tile(pl)
        out = torch.zeros(m, ld, dtype=torch.float32 if o32 else torch.float16, device="cuda")
        ops.gemm_nt(a, w, out, bias=bias, residual=res, epilogue=epi, n=n)
        outs[pl] = out
    torch.cuda.synchronize()
    ref = outs[256] if 256 in outs else outs[plans[0]]
    equal = {pl: bool(torch.equal(outs[pl][:, :n], ref[:, :n])) for pl in plans}
    maxdiff = {pl: float((outs[pl][:, :n].float() - ref[:, :n].float()).abs().max()) for pl in plans}
    # spot check against fp32 torch on a row sample
    rows = torch.arange(0, m, max(1, m // 64), device="cuda")[:64]
    exact = a[rows].float() @ w.float().t()
    if bias is not None:
        exact = exact + bias
    if epi == EPI_BIAS_QUICKGELU:
        exact = exact * torch.sigmoid(1.702 * exact)
    if res is not None:
        exact = exact + res[rows, :n].float()
    if epi in (4, 5):
        exact = torch.relu(exact)
    err = {pl: float((outs[pl][rows, :n].float() - exact).abs().max()) for pl in plans}

    def make(pl):
        out = torch.zeros(m, ld, dtype=torch.float32 if o32 else torch.float16, device="cuda")
        r = out if epi == EPI_BIAS_RESIDUAL else res       # fp32 residual in place, as the towers call it; 16-bit identity as is

        def f():
            ops.gemm_set_tile(pl)
            ops.gemm_nt(a, w, out, bias=bias, residual=r, epilogue=epi, n=n)
        return f

    fns = {pl: make(pl) for pl in plans}
    for f in fns.values():
        f(); f()
    ts = {pl: [] for pl in plans}
    for rep in range(7):
        for pl, f in fns.items():
            ts[pl].append(timeit(f))
    line = {"name": name, "shape": [m, n, k], "equal_to_256": equal, "max_diff": maxdiff, "err_vs_fp32": err}
    for pl in plans:
        t = sorted(ts[pl])
        line[f"us_{pl}"] = [round(t[0], 1), round(t[len(t) // 2], 1)]
        line[f"tf_{pl}"] = round(2.0 * m * n * k / t[0] / 1e6)
    if "--blas" in sys.argv or True:
        c = torch.empty(m, n, dtype=torch.float16, device="cuda")
        wt = w.t()
        fb = lambda: torch.matmul(a, wt, out=c)
        fb(); fb()
        tb = sorted(timeit(fb) for _ in range(5))
        line["us_hipblaslt_plain"] = round(tb[0], 1)
    print(json.dumps(line), flush=True)
ops.gemm_set_tile(0)
