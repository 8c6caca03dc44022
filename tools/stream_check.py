#!/usr/bin/env python3
"""Dev tool: exactness of hgr_gemm_nt's few-rows x many-columns paths on small-integer operands (every sum exact in fp32),
ragged shapes, padded / unpadded ldc, nothing written past N.  Run once per HGR_GEMM_STREAM setting."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hgr_net_amd import ops
ok = True
for dt in (torch.float16, torch.bfloat16):
    for m, n, k in [(512, 21841, 512), (200, 5000, 128), (500, 4099, 384), (1, 4096, 128), (300, 9000, 256), (64, 24000, 768), (512, 20842, 512)]:
        gen = torch.Generator().manual_seed(m + n + k)
        a = torch.randint(-2, 3, (m, k), generator=gen).float()
        w = torch.randint(-1, 2, (n, k), generator=gen).float()
        ad, wd = a.to(dt).cuda(), w.to(dt).cuda()
        base = ad.float() @ wd.float().t()
        for ldc in sorted({(n + 3) // 4 * 4, (n + 63) // 64 * 64 + 64}):
            out = torch.full((m, ldc), 7.0, dtype=torch.float32, device="cuda")
            ops.gemm_nt(ad, wd, out, n=n)
            good = bool(torch.equal(out[:, :n], base)) and bool((out[:, n:] == 7.0).all())
            ok &= good
            if not good:
                bad = (out[:, :n] != base).nonzero()
                print("MISMATCH", dt, m, n, k, ldc, bad[:5].tolist(), "pad ok" if bool((out[:, n:] == 7.0).all()) else "pad written")
print("stream_check", os.environ.get("HGR_GEMM_STREAM", "1"), "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
