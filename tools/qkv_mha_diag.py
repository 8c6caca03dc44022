#!/usr/bin/env python3
"""Dev tool: where do hgr_gemm_nt_ln_mha and hgr_gemm_nt_ln + hgr_mha differ (f16, ViT-B/32 shape)?  Prints the mismatch pattern."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from hgr_net_amd import ops, synth

DEV = "cuda"
dt = torch.float16
b, l, heads = 512, 50, 12
w, m = heads * 64, b * l


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((scale * synth.normal(seed, "t", int(np.prod(shape)))).astype(np.float32).reshape(shape))


x = (_rand((m, w), 41, 1.2) + 0.2 * _rand((m, 1), 42))
x = x * (0.6 + torch.rand(m, 1, generator=torch.Generator().manual_seed(9)))
w_in, b_in = _rand((3 * w, w), 43, w ** -0.5), 0.1 * _rand((3 * w,), 44)
gamma, beta = 1.0 + 0.2 * _rand((w,), 45), 0.1 * _rand((w,), 46)
xd = x.to(DEV)
x16 = torch.empty(m, w, dtype=dt, device=DEV)
xlo = torch.empty(m, w, dtype=torch.float16, device=DEV)
stats = torch.empty(m, w // 64, 2, dtype=torch.float32, device=DEV)
ops.row_stats16(xd, x16, xlo, stats)
wf = (w_in * gamma[None, :]).to(dt)
s = wf.float().sum(1).to(DEV)
c = (w_in @ beta + b_in).to(DEV)
wf = wf.to(DEV)
qkv = torch.empty(m, 3 * w, dtype=dt, device=DEV)
want = torch.empty(m, w, dtype=dt, device=DEV)
ops.gemm_nt_ln(x16, wf, qkv, s, c, stats, 1e-5)
ops.mha(qkv, want, b, l, heads, False)
got = torch.empty(m, w, dtype=dt, device=DEV)
ops.gemm_nt_ln_mha(x16, wf, got, s, c, stats, b, l, heads, False, 1e-5)
torch.cuda.synchronize()
bad = (got != want).nonzero().cpu().numpy()
print("mismatches", len(bad))
rows, cols = bad[:, 0], bad[:, 1]
seq, tok, head, dim = rows // l, rows % l, cols // 64, cols % 64
tile, trow = rows // 250, rows % 250
print("distinct (seq, head):", len(set(zip(seq.tolist(), head.tolist()))), " distinct rows:", len(set(rows.tolist())), " distinct seqs:", len(set(seq.tolist())))
print("per query-tile index inside the 256-row tile (trow // 16):", np.bincount(trow // 16, minlength=16).tolist())
print("per token position:", np.bincount(tok, minlength=l).tolist())
print("per head:", np.bincount(head, minlength=heads).tolist())
print("per dim // 16:", np.bincount(dim // 16, minlength=4).tolist())
from collections import Counter
cnt = Counter(zip(seq.tolist(), head.tolist()))
print("largest (seq, head) groups:", cnt.most_common(8))
# second opinion on q/k/v: attention of the qkv buffer by torch in fp32, against both
q, k_, v = qkv.float().view(b, l, 3, heads, 64).unbind(2)
att = torch.softmax(torch.einsum("blhd,bmhd->bhlm", q, k_) * 0.125, dim=-1)
ref = torch.einsum("bhlm,bmhd->blhd", att, v).reshape(m, w)
print("max |fused - fp32 ref|", float((got.float() - ref).abs().max()), " max |unfused - fp32 ref|", float((want.float() - ref).abs().max()))
e1 = (got.float() - ref).abs()[rows, cols].mean().item()
e2 = (want.float() - ref).abs()[rows, cols].mean().item()
print("mean |err| on the mismatching elements: fused", e1, " unfused", e2)
