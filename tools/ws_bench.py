#!/usr/bin/env python3
"""Dev tool (round 5): the role-split GEMM (gemm_nt_ws, csrc/hgr_gemm_ws.hip) against gemm_nt_duo through the public entry points
(hgr_gemm_nt, hgr_gemm_nt_ln, hgr_gemm_nt_res_stats_guard), switched per call with hgr_gemm_set_ws.  Correctness first (bit
identity on multi-tile shapes), then interleaved timing rounds in one process."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops, _lib
from hgr_net_amd._lib import EPI_NONE, EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RELU


def set_ws(on):
    return _lib.load().hgr_gemm_set_ws(1 if on else 0)


def mk(m, n, k, dt=torch.float16, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = ((torch.rand(m, k, device="cuda", generator=g) * 2 - 1)).to(dt)
    w = ((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * 0.05).to(dt)
    return a, w, g


class Plain:
    def __init__(self, m, n, k, epi, dt=torch.float16):
        self.a, self.w, g = mk(m, n, k, dt)
        self.bias = torch.rand(n, device="cuda", generator=g) - 0.5 if epi != EPI_NONE else None
        self.epi, self.flops = epi, 2.0 * m * n * k
        self.out = [torch.full((m, n), float("nan"), dtype=dt, device="cuda") for _ in range(2)]

    def run(self, ws):
        set_ws(ws)
        ops.gemm_nt(self.a, self.w, self.out[ws], bias=self.bias, epilogue=self.epi)

    def same(self):
        return bool(torch.equal(self.out[0].view(torch.int16), self.out[1].view(torch.int16)))


class LnC:
    def __init__(self, m, n, k, act, dt=torch.float16):
        self.a, self.w, g = mk(m, n, k, dt)
        self.s = torch.rand(n, device="cuda", generator=g) - 0.5
        self.c = torch.rand(n, device="cuda", generator=g) - 0.5
        x = self.a.float().view(m, k // 64, 64)
        self.stats = torch.stack([x.sum(-1), (x * x).sum(-1)], dim=-1).contiguous()
        self.act, self.flops = act, 2.0 * m * n * k
        self.out = [torch.full((m, n), float("nan"), dtype=dt, device="cuda") for _ in range(2)]

    def run(self, ws):
        set_ws(ws)
        ops.gemm_nt_ln(self.a, self.w, self.out[ws], self.s, self.c, self.stats, quickgelu=self.act)

    def same(self):
        return bool(torch.equal(self.out[0].view(torch.int16), self.out[1].view(torch.int16)))


class LnP:
    def __init__(self, m, n, k, dt=torch.float16):
        self.a, self.w, g = mk(m, n, k, dt)
        self.bias = torch.rand(n, device="cuda", generator=g) - 0.5
        self.xh0 = (torch.rand(m, n, device="cuda", generator=g) * 4 - 2).to(dt)
        self.xl0 = torch.randint(0, 256, (m, n), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
        self.flops = 2.0 * m * n * k
        self.xh = [torch.empty_like(self.xh0) for _ in range(2)]
        self.xl = [torch.empty_like(self.xl0) for _ in range(2)]
        self.st = [torch.full((m, n // 64, 2), float("nan"), device="cuda") for _ in range(2)]
        self.flag = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(2)]
        self.fresh = True

    def reset(self):
        for i in range(2):
            self.xh[i].copy_(self.xh0); self.xl[i].copy_(self.xl0)

    def run(self, ws):
        set_ws(ws)
        ops.gemm_nt_res_stats(self.a, self.w, self.xh[ws], self.xl[ws], self.bias, self.st[ws], flag=self.flag[ws])

    def same(self):
        return bool(torch.equal(self.xh[0].view(torch.int16), self.xh[1].view(torch.int16)) and torch.equal(self.xl[0], self.xl[1])
                    and torch.equal(self.st[0].view(torch.int32), self.st[1].view(torch.int32)) and int(self.flag[0]) == int(self.flag[1]))


def check(name, case):
    if hasattr(case, "reset"):
        case.reset()
    case.run(0); case.run(1)
    torch.cuda.synchronize()
    ok = case.same()
    print(json.dumps({"check": name, "bit_identical": ok}), flush=True)
    return ok


def timeit(f, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def bench(name, case, rounds=6, iters=20):
    for _ in range(3):
        case.run(0); case.run(1)
    torch.cuda.synchronize()
    t = [[], []]
    for _ in range(rounds):
        for ws in (0, 1):
            t[ws].append(timeit(lambda: case.run(ws), iters))
    r = {"shape": name, "duo_us": [round(min(t[0]), 1), round(sorted(t[0])[len(t[0]) // 2], 1)], "ws_us": [round(min(t[1]), 1), round(sorted(t[1])[len(t[1]) // 2], 1)],
         "duo_tf": round(case.flops / min(t[0]) / 1e6), "ws_tf": round(case.flops / min(t[1]) / 1e6)}
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    ok = True
    M = 25600
    for dt in (torch.float16, torch.bfloat16):
        tag = "f16" if dt == torch.float16 else "bf16"
        ok &= check(f"plain none {tag} 4096x4096x512", Plain(4096, 4096, 512, EPI_NONE, dt))
        ok &= check(f"plain bias {tag} 8192x2048x512", Plain(8192, 2048, 512, EPI_BIAS, dt))
        ok &= check(f"plain bias {tag} fc", Plain(M, 3072, 768, EPI_BIAS, dt))
        ok &= check(f"plain relu {tag} 8192x1024x1024", Plain(8192, 1024, 1024, EPI_BIAS_RELU, dt))
        ok &= check(f"lnc gelu {tag} fc", LnC(M, 3072, 768, True, dt))
        ok &= check(f"lnc {tag} qkv", LnC(M, 2304, 768, False, dt))
        ok &= check(f"lnc {tag} 16384x1024x512", LnC(16384, 1024, 512, False, dt))
        ok &= check(f"lnc gelu {tag} 8192x4096x1024", LnC(8192, 4096, 1024, True, dt))
        ok &= check(f"lnp {tag} out", LnP(M, 768, 768, dt))
        ok &= check(f"lnp {tag} proj", LnP(M, 768, 3072, dt))
        ok &= check(f"lnp {tag} 16384x512x512", LnP(16384, 512, 512, dt))
    if not ok and "--force" not in sys.argv:
        print("MISMATCH - not timing"); sys.exit(1)
    bench("fc lnc+gelu", LnC(M, 3072, 768, True))
    bench("qkv lnc", LnC(M, 2304, 768, False))
    bench("out lnp", LnP(M, 768, 768))
    bench("proj lnp", LnP(M, 768, 3072))
    bench("patch plain-none 25088x768x3072", Plain(25088, 768, 3072, EPI_NONE))
    bench("fc plain bias", Plain(M, 3072, 768, EPI_BIAS))
    set_ws(1)
