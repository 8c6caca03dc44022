#!/usr/bin/env python3
"""Dev tool: time one GEMM shape under HGR_GEMM_DBG / HGR_GEMM_TILE settings (run once per setting).
(round 6: these switches exist in the lab build only - `make -C hgr-net_amd/csrc lab`, then HGR_LIB=$PWD/hgr-net_amd/lib/libhgr_lab.so)"""
import sys, os, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_NONE, EPI_BIAS, EPI_BIAS_QUICKGELU
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
res = {}
for name, m, n, k, epi in [("fc", 25600, 3072, 768, EPI_BIAS_QUICKGELU), ("qkv", 25600, 2304, 768, EPI_BIAS), ("sq4096", 4096, 4096, 4096, EPI_NONE), ("k8192", 4096, 4096, 8192, EPI_NONE)]:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).half(); w = ((torch.rand(n, k, device="cuda") * 2 - 1) * 0.05).half()
    bias = torch.rand(n, device="cuda"); out = torch.empty(m, n, dtype=torch.float16, device="cuda")
    us = timeit(lambda: ops.gemm_nt(a, w, out, bias=bias if epi else None, epilogue=epi))
    res[name] = (round(us, 1), round(2.0 * m * n * k / us / 1e6, 0))
print(os.environ.get("HGR_GEMM_DBG", "0"), os.environ.get("HGR_GEMM_TILE", "auto"), json.dumps(res))
