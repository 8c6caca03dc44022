#!/usr/bin/env python3
"""Dev tool: the weight-gradient GEMM (hgr_gemm_tn_splitk: partial[s] = P[slice s]^T . Q[slice s]) of ONE build of libhgr.so on the
shapes of the ViT-L/14 + CoOp training step (configs[4]: 256 images x 257 tokens = 65792 rows, width 1024; text 300 prompts x 77 = 23100
rows, width 768), bound through plain ctypes so that builds compare across ABI changes:

    HGR_LIB=/path/to/libhgr.so python tools/tn_ab.py             (alternate the builds on one box: tools/lib_ab.sh tools/tn_ab.py)

Back-to-back launches over rotating operand sets, min / median of 5 rounds x 6 launches; us and TFLOP/s."""
import ctypes as C
import json
import os
from pathlib import Path

import torch

lib_path = os.environ.get("HGR_LIB", str(Path(__file__).resolve().parent.parent / "hgr-net_amd" / "lib" / "libhgr.so"))
lib = C.CDLL(lib_path)
p, i, l = C.c_void_p, C.c_int, C.c_int64
lib.hgr_gemm_tn_splitk.argtypes = [p, l, p, l, p, l, i, i, i, i, i, p]
lib.hgr_gemm_tn_tile.argtypes = [i, i]
lib.hgr_last_error.restype = C.c_char_p

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(1)
st = lambda: torch.cuda.current_stream().cuda_stream


def slices(na, nb, m):
    t = lib.hgr_gemm_tn_tile(na, nb)
    tiles, slots = -(-na // t) * -(-nb // t), 256 if t == 256 else 512
    best = (0.0, 1)
    for s in range(1, max(1, min(m // 512, 4096)) + 1):
        wgs = tiles * s
        eff = wgs / (-(-wgs // slots) * slots)
        if eff >= 0.85:
            return s
        if eff > best[0]:
            best = (eff, s)
    return best[1]


def timeit(fn, iters=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for k in range(iters):
        fn(k)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


res = {}
for name, m, na, nb in (("v_qkv", 65792, 1024, 3072), ("v_out", 65792, 1024, 1024), ("v_fc", 65792, 1024, 4096), ("v_proj", 65792, 4096, 1024),
                        ("t_fc", 23100, 768, 3072), ("t_qkv", 23100, 768, 2304)):
    NB = 3
    P = [torch.randn((m, na), generator=g, device=DEV).half() for _ in range(NB)]
    Q = [torch.randn((m, nb), generator=g, device=DEV).half() for _ in range(NB)]
    s = slices(na, nb, m)
    kc = -(-(-(-m // s)) // 64) * 64
    s = -(-m // kc)
    part = torch.empty((s, na, nb), dtype=torch.float32, device=DEV)

    def fn(k):
        if lib.hgr_gemm_tn_splitk(P[k % NB].data_ptr(), na, Q[k % NB].data_ptr(), nb, part.data_ptr(), nb, m, na, nb, kc, 1, st()):
            raise RuntimeError(lib.hgr_last_error().decode())

    fn(0)
    ts = sorted(timeit(fn) for _ in range(5))
    chk = float(part[:, :: max(1, na // 7), :: max(1, nb // 5)].double().sum())
    res[name] = {"us_min": round(ts[0], 1), "us_med": round(ts[2], 1), "tflops": round(2.0 * m * na * nb / ts[0] / 1e6, 1), "slices": s, "chk": chk}
    del P, Q, part
print(json.dumps({"lib": os.path.basename(lib_path), "tile": lib.hgr_gemm_tn_tile(1024, 1024), "gemm_tn": res}))
