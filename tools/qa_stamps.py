#!/usr/bin/env python3
"""Dev tool (round 6, lab build only): where a tile of qkv_attn (in_proj GEMM + attention, csrc/hgr_qkv_attn.hip) spends its time -
s_memtime stamps of wave 0 around the sections of every workgroup's second tile, read back with hgr_lab_qa_stamps.
    HGR_LIB=$PWD/hgr-net_amd/lib/libhgr_lab.so python tools/qa_stamps.py        (make -C hgr-net_amd/csrc lab)"""
import sys, json, ctypes
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from hgr_net_amd import ops, _lib

lib = _lib.load()
b, l, heads = 512, 50, 12
w = heads * 64
m = b * l
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand(m, w, device="cuda", generator=g) * 2 - 1).half()
wf = ((torch.rand(3 * w, w, device="cuda", generator=g) * 2 - 1) * 0.05).half()
s = torch.rand(3 * w, device="cuda", generator=g) - 0.5
c = torch.rand(3 * w, device="cuda", generator=g) - 0.5
xs = x.float().view(m, w // 64, 64)
stats = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], dim=-1).contiguous()
att = torch.empty(m, w, dtype=torch.float16, device="cuda")
for _ in range(20):
    ops.gemm_nt_ln_mha(x, wf, att, s, c, stats, b, l, heads, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm_nt_ln_mha(x, wf, att, s, c, stats, b, l, heads, False)
e1.record()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8))()
rc = lib.hgr_lab_qa_stamps(buf, 256 * 8)
assert rc == 0, rc
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
d = np.diff(st[:, :6], axis=1)
names = ["main loop", "row statistics -> LDS", "q/k/v -> LDS (+ next tile's request)", "attention", "seam barrier"]
out = {"launch_us": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "clock_note": "ticks = s_memtime; ticks_per_us against s_memrealtime (100 MHz) tells what a tick is"}
for i, n in enumerate(names):
    out[n] = {"median_ticks": int(np.median(d[:, i])), "p10": int(np.percentile(d[:, i], 10)), "p90": int(np.percentile(d[:, i], 90))}
out["tile_total_ticks_median"] = int(np.median(st[:, 5] - st[:, 0]))
rt = (st[:, 7] - st[:, 6]).astype(np.float64)                      # s_memrealtime: 100 MHz
ok = rt > 0
out["tile_total_us_median"] = round(float(np.median(rt[ok])) / 100.0, 2)
out["ticks_per_us"] = round(float(np.median((st[:, 5] - st[:, 0])[ok] / (rt[ok] / 100.0))), 1)
print(json.dumps(out, indent=1))
