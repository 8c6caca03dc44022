#!/usr/bin/env python3
"""Dev tool: the LayerNorm-producer GEMM (hgr_gemm_nt_res_stats: out_proj / c_proj of ViT-B/32 at batch 512) and the consumer
(hgr_gemm_nt_ln: in_proj / c_fc) of ONE build of libhgr.so, bound through plain ctypes (only entry points every build since round 2
exports), so that builds can be compared across ABI changes:

    HGR_LIB=/path/to/libhgr.so python tools/producer_ab.py            (run the builds alternately on one box: tools/lib_ab.sh)

Back-to-back launches over rotating operand sets (activations from beyond the L2), min / median of 5 rounds x 12 launches, us."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import torch

lib_path = os.environ.get("HGR_LIB", str(Path(__file__).resolve().parent.parent / "hgr-net_amd" / "lib" / "libhgr.so"))
lib = C.CDLL(lib_path)
p, i, l, f = C.c_void_p, C.c_int, C.c_int64, C.c_float
lib.hgr_gemm_nt_res_stats.argtypes = [p, l, p, l, p, p, l, p, p, i, i, i, i, p]
lib.hgr_gemm_nt_ln.argtypes = [p, l, p, l, p, l, p, p, p, f, i, i, i, i, i, p]
lib.hgr_last_error.restype = C.c_char_p
has_tail = hasattr(lib, "hgr_gemm_set_tail")
if has_tail:
    lib.hgr_gemm_set_tail.argtypes = [i, i]

DEV, M = "cuda", 25600
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda shape, scale=1.0: scale * torch.randn(shape, generator=g, device=DEV)
NB = 4
x768 = [rnd((M, 768)).half() for _ in range(NB)]
u3072 = [rnd((M, 3072)).half() for _ in range(NB)]
xh = [rnd((M, 768)).half() for _ in range(NB)]
xl = [torch.full((M, 768), 128, dtype=torch.uint8, device="cuda") for _ in range(NB)]     # the pair's low byte (round 4; libraries of rounds 2 - 3 read it as half of an f16 plane: timings stay comparable, values do not)
stats = torch.rand((M, 12, 2), device=DEV) + 700.0           # plausible (sum, sum of squares) so that rstd is finite
w_out, w_proj = rnd((768, 768), 0.03).half(), rnd((768, 3072), 0.03).half()
w_qkv, w_fc = rnd((2304, 768), 0.03).half(), rnd((3072, 768), 0.03).half()
b768, s_q, c_q, s_f, c_f = rnd((768,)), rnd((2304,)), rnd((2304,)), rnd((3072,)), rnd((3072,))
qkv = [torch.empty((M, 2304), dtype=torch.float16, device=DEV) for _ in range(2)]
u = [torch.empty((M, 3072), dtype=torch.float16, device=DEV) for _ in range(2)]
st = lambda: torch.cuda.current_stream().cuda_stream


def chk(rc):
    if rc:
        raise RuntimeError(lib.hgr_last_error().decode())


def out(k):
    chk(lib.hgr_gemm_nt_res_stats(x768[k % NB].data_ptr(), 768, w_out.data_ptr(), 768, xh[k % NB].data_ptr(), xl[k % NB].data_ptr(), 768, b768.data_ptr(),
                                  stats.data_ptr(), M, 768, 768, 1, st()))


def proj(k):
    chk(lib.hgr_gemm_nt_res_stats(u3072[k % NB].data_ptr(), 3072, w_proj.data_ptr(), 3072, xh[k % NB].data_ptr(), xl[k % NB].data_ptr(), 768, b768.data_ptr(),
                                  stats.data_ptr(), M, 768, 3072, 1, st()))


def qk(k):
    chk(lib.hgr_gemm_nt_ln(x768[k % NB].data_ptr(), 768, w_qkv.data_ptr(), 768, qkv[k & 1].data_ptr(), 2304, s_q.data_ptr(), c_q.data_ptr(), stats.data_ptr(),
                           1e-5, M, 2304, 768, 1, 0, st()))


def fc(k):
    chk(lib.hgr_gemm_nt_ln(x768[k % NB].data_ptr(), 768, w_fc.data_ptr(), 768, u[k & 1].data_ptr(), 3072, s_f.data_ptr(), c_f.data_ptr(), stats.data_ptr(),
                           1e-5, M, 3072, 768, 1, 1, st()))


def timeit(fn, iters=12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(iters):
        fn(k)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
arms = [("", None, -1)] if not has_tail else [("tail_off", 0, -1), ("tail_on", 1, -1)]
if has_tail and os.environ.get("AB_PB"):                      # forced full-panel counts, e.g. AB_PB=80,72,64
    arms += [(f"pb{v}", 1, int(v)) for v in os.environ["AB_PB"].split(",")]
for _ in range(40):                                           # ~0.2 s of launches first: the first arms otherwise run on ramping clocks
    timeit(fc)
for name, fn in (("out", out), ("proj", proj), ("qkv", qk), ("fc", fc)):
    ts = {tag: [] for tag, _, _ in arms}
    for _ in range(5):                                        # arms interleaved (CDNA guide rule 24)
        for tag, en, pb in arms:
            if en is not None:
                lib.hgr_gemm_set_tail(en, pb)
            fn(0)
            ts[tag].append(timeit(fn))
    for tag, v in ts.items():
        v.sort()
        res[name + ("_" + tag if tag else "")] = [round(v[0], 1), round(v[2], 1)]
print(json.dumps({"lib": os.path.basename(lib_path), "us_min_med": res}))
