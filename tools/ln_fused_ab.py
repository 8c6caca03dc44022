#!/usr/bin/env python3
"""Dev tool: the LayerNorm-folded GEMM pair against the unfused launches it replaces, same process, interleaved
(ViT-B/32 shapes at batch 512): out-proj / c_proj with and without the 16-bit copy + slot statistics, in_proj / c_fc with
the folded LayerNorm against LayerNorm kernel + GEMM."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL

dt = torch.float16
m, w = 25600, 768


def timeit(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def ab(name, fns, reps=7):
    for f in fns.values():
        f(); f()
    ts = {k: [] for k in fns}
    for _ in range(reps):
        for k, f in fns.items():
            ts[k].append(timeit(f))
    print(json.dumps({"name": name, **{k: [round(min(v), 1), round(sorted(v)[len(v) // 2], 1)] for k, v in ts.items()}}), flush=True)


torch.manual_seed(0)
x = torch.randn(m, w, device="cuda")
x16 = torch.empty(m, w, dtype=dt, device="cuda")
xlo = torch.empty(m, w, dtype=torch.float16, device="cuda")
stats = torch.empty(m, w // 64, 2, device="cuda")
h16 = torch.empty(m, w, dtype=dt, device="cuda")
gamma, beta = torch.rand(w, device="cuda") + 0.5, torch.randn(w, device="cuda") * 0.1
for name, k_in, n_out, gelu in (("qkv", w, 3 * w, False), ("fc", w, 4 * w, True)):
    wt = (torch.randn(n_out, k_in, device="cuda") * 0.03)
    b = torch.randn(n_out, device="cuda")
    wf = (wt * gamma[None]).to(dt)
    s, c = wf.float().sum(1).contiguous(), (wt @ beta + b).contiguous()
    w16 = wt.to(dt)
    out = torch.empty(m, n_out, dtype=dt, device="cuda")
    ops.row_stats16(x, x16, xlo, stats)
    ab(name, {"ln_kernel": lambda: ops.layernorm(x, gamma, beta, h16),
              "gemm_bias": lambda: ops.gemm_nt(h16, w16, out, bias=b, epilogue=EPI_BIAS_QUICKGELU if gelu else EPI_BIAS),
              "gemm_ln_folded": lambda: ops.gemm_nt_ln(x16, wf, out, s, c, stats, 1e-5, quickgelu=gelu)})
for name, k_in in (("out", w), ("proj", 4 * w)):
    a = (torch.randn(m, k_in, device="cuda") * 0.5).to(dt)
    wt = (torch.randn(w, k_in, device="cuda") * 0.03).to(dt)
    b = torch.randn(w, device="cuda")
    xr = torch.randn(m, w, device="cuda")
    ab(name, {"gemm_residual": lambda: ops.gemm_nt(a, wt, xr, bias=b, residual=xr, epilogue=EPI_BIAS_RESIDUAL),
              "gemm_res_stats_pair": lambda: ops.gemm_nt_res_stats(a, wt, x16, xlo, b, stats)})
