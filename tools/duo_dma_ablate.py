#!/usr/bin/env python3
"""Dev tool (round 5): gemm_nt_duo with parts of its LDS-DMA stream left out (HGR_GEMM_DBG = 128: no A pieces of odd K-tiles,
256: no A pieces, 512: no W pieces, behind the first two K-tiles; wrong results) - what the operand fill costs the two-workgroup form.
One process per setting.
(round 6: these switches exist in the lab build only - `make -C hgr-net_amd/csrc lab`, then HGR_LIB=$PWD/hgr-net_amd/lib/libhgr_lab.so)"""
import os, sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from ws_bench import LnC, LnP, timeit
res = {}
for name, case in (("fc_lnc_gelu", LnC(25600, 3072, 768, True)), ("proj_lnp", LnP(25600, 768, 3072)), ("out_lnp", LnP(25600, 768, 768))):
    f = lambda: case.run(0)
    for _ in range(5): f()
    torch.cuda.synchronize()
    res[name] = round(min(timeit(f, 20) for _ in range(5)), 1)
print("HGR_GEMM_DBG=" + os.environ.get("HGR_GEMM_DBG", "0"), json.dumps(res), flush=True)
