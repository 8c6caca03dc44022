#!/usr/bin/env python3
"""Dev tool (round 5): time the bare product of gemm_nt_ws under HGR_WS_DBG (one process per setting; wrong results by design for
DBG != 0: 1 no MFMAs, 2 no LDS-DMAs, 4 no fragment reads, 8 no epilogue stores; sums combine).
(round 6: these switches exist in the lab build only - `make -C hgr-net_amd/csrc lab`, then HGR_LIB=$PWD/hgr-net_amd/lib/libhgr_lab.so)"""
import os, sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_bench import Plain, timeit
from hgr_net_amd._lib import EPI_NONE
res = {}
for name, m, n, k in [("fc", 25600, 3072, 768), ("proj", 25600, 768, 3072), ("sq4096", 4096, 4096, 4096)]:
    c = Plain(m, n, k, EPI_NONE)
    f = lambda: c.run(1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    t = min(timeit(f, 20) for _ in range(5))
    res[name] = [round(t, 1), round(2.0 * m * n * k / t / 1e6)]
print("HGR_WS_DBG=" + os.environ.get("HGR_WS_DBG", "0"), json.dumps(res), flush=True)
