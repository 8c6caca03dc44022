#!/usr/bin/env python3
"""Dev tool (round 5): the persistent 256 x 256 consumer GEMM (gemm_nt_p8, csrc/hgr_gemm_p8.hip) against gemm_nt_duo through
hgr_gemm_nt_ln, switched per call with hgr_gemm_set_p8: bit identity first, then interleaved timing rounds in one process."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops, _lib
from tools.ws_bench import LnC, timeit


def set_p8(on):
    return _lib.load().hgr_gemm_set_p8(1 if on else 0)  # 1 = wherever it covers, 0 = never


class Case(LnC):
    def run(self, on):
        set_p8(on)
        ops.gemm_nt_ln(self.a, self.w, self.out[on], self.s, self.c, self.stats, quickgelu=self.act)


def check(name, c):
    c.run(0); c.run(1)
    torch.cuda.synchronize()
    same = c.same()
    d = (c.out[0].float() - c.out[1].float()).abs()
    print(json.dumps({"check": name, "bit_identical": same, "max_abs_diff": float(d.max()), "n_diff": int((d > 0).sum()), "nan": bool(torch.isnan(c.out[1].float()).any())}), flush=True)
    return same


def bench(name, c, rounds=6, iters=20):
    for _ in range(3):
        c.run(0); c.run(1)
    torch.cuda.synchronize()
    t = [[], []]
    for _ in range(rounds):
        for on in (0, 1):
            t[on].append(timeit(lambda: c.run(on), iters))
    print(json.dumps({"shape": name, "duo_us": round(min(t[0]), 1), "p8_us": round(min(t[1]), 1),
                      "duo_tf": round(c.flops / min(t[0]) / 1e6), "p8_tf": round(c.flops / min(t[1]) / 1e6)}), flush=True)


if __name__ == "__main__":
    M = 25600
    ok = True
    for dt in (torch.float16, torch.bfloat16):
        tag = "f16" if dt == torch.float16 else "bf16"
        ok &= check(f"gelu {tag} fc", Case(M, 3072, 768, True, dt))
        ok &= check(f"plain {tag} 16384x4096x512", Case(16384, 4096, 512, False, dt))
        ok &= check(f"gelu {tag} 65536x1024x1024 (ln_slots 16)", Case(65536, 1024, 1024, True, dt))
    if not ok and "--force" not in sys.argv:
        print("MISMATCH - not timing"); sys.exit(1)
    bench("fc lnc+gelu 25600x3072x768", Case(M, 3072, 768, True))
    bench("qkv-like 25600x2304x768", Case(M, 2304, 768, False))
    bench("25600x3072x3072", Case(M, 3072, 3072, False))
    bench("ViT-L fc 65792->65536x4096x1024", Case(65536, 4096, 1024, True, torch.bfloat16))
    _lib.load().hgr_gemm_set_p8(2)
