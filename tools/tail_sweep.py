#!/usr/bin/env python3
"""Dev tool: the tail plan of gemm_nt_duo (half tiles on the last row panels, hgr_gemm_set_tail) on the ViT-B/32 tower launches at
batch 512, through the entry points the tower uses (LayerNorm consumer for qkv / fc, producer with the pair read-modify-write and
slot statistics for out / proj, plain fp32 for the patch GEMM), ONE process, interleaved rounds (CDNA guide rule 24).

    tail_sweep.py [--sweep]      per shape: tail off / auto, and with --sweep forced full-panel counts around the model's choice

Operands rotate over several buffers larger than the Infinity Cache so that launches see what the step hands them (activations
from HBM / MALL, not from a warm L2)."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import ops

DEV = "cuda"
DT = torch.float16
M = 25600


def ev_time(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def shapes():
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda shape, scale=1.0: scale * torch.randn(shape, generator=g, device=DEV)
    NB = 4                                                     # operand sets in rotation
    x768 = [rnd((M, 768)).to(DT) for _ in range(NB)]
    u3072 = [rnd((M, 3072)).to(DT) for _ in range(NB)]
    xh = [rnd((M, 768)).to(DT) for _ in range(NB)]
    xl = [torch.full((M, 768), 128, dtype=torch.uint8, device="cuda") for _ in range(NB)]
    stats = torch.empty((M, 12, 2), dtype=torch.float32, device=DEV)
    ops.row_stats16(rnd((M, 768)), xh[0].clone(), xl[0].clone(), stats)
    w_qkv, w_out, w_fc, w_proj = rnd((2304, 768), 0.03).to(DT), rnd((768, 768), 0.03).to(DT), rnd((3072, 768), 0.03).to(DT), rnd((768, 3072), 0.03).to(DT)
    s_q, c_q, s_f, c_f, b768 = rnd((2304,)), rnd((2304,)), rnd((3072,)), rnd((3072,)), rnd((768,))
    qkv = [torch.empty((M, 2304), dtype=DT, device=DEV) for _ in range(2)]
    u = [torch.empty((M, 3072), dtype=DT, device=DEV) for _ in range(2)]
    pe = torch.empty((25088, 768), dtype=torch.float32, device=DEV)
    patches = [rnd((25088, 3072)).to(DT) for _ in range(2)]
    w_patch = rnd((768, 3072), 0.03).to(DT)
    return [
        ("qkv", 2304, lambda i: ops.gemm_nt_ln(x768[i % NB], w_qkv, qkv[i & 1], s_q, c_q, stats, 1e-5)),
        ("out", 768, lambda i: ops.gemm_nt_res_stats(x768[i % NB], w_out, xh[i % NB], xl[i % NB], b768, stats)),
        ("fc", 3072, lambda i: ops.gemm_nt_ln(x768[i % NB], w_fc, u[i & 1], s_f, c_f, stats, 1e-5, quickgelu=True)),
        ("proj", 768, lambda i: ops.gemm_nt_res_stats(u3072[i % NB], w_proj, xh[i % NB], xl[i % NB], b768, stats)),
        ("patch", 768, lambda i: ops.gemm_nt(patches[i & 1], w_patch, pe)),
    ]


def main():
    sweep = "--sweep" in sys.argv
    rounds, iters = 5, 12
    res = {}
    for name, n, fn in shapes():
        tiles_m, tiles_n = (M + 255) // 256, n // 128
        arms = [("off", (False, -1)), ("auto", (True, -1))]
        if sweep:
            full_rounds = (tiles_m * tiles_n) // 512
            base = (full_rounds * 512) // tiles_n
            for pb in sorted({0, base // 2, max(0, base - 8), max(0, base - 2), base - 1, base, base + 1}):
                if 0 <= pb < tiles_m:
                    arms.append((f"pb{pb}", (True, pb)))
        best = {a: [] for a, _ in arms}
        for _ in range(rounds):
            for a, (en, pb) in arms:
                ops.gemm_set_tail(en, pb)
                fn(0)
                best[a].append(ev_time(fn, iters))
        ops.gemm_set_tail(True, -1)
        res[name] = {a: {"min": round(min(v), 1), "med": round(sorted(v)[len(v) // 2], 1)} for a, v in best.items()}
        print(name, json.dumps(res[name]), flush=True)
    print(json.dumps({"tail_sweep": res}))


if __name__ == "__main__":
    main()
