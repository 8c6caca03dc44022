#!/usr/bin/env python3
"""Dev tool: A/B two builds of libhgr's hgr_mha in ONE process (interleaved), outputs compared bit for bit.
usage: mha_ab.py old.so new.so"""
import sys, ctypes as C, json
import torch
_p, _i = C.c_void_p, C.c_int
libs = []
for path in sys.argv[1:3]:
    lib = C.CDLL(path); lib.hgr_mha.argtypes = [_p, _p, _i, _i, _i, _i, _i, _p]; lib.hgr_mha.restype = _i; libs.append(lib)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
# (name, B, L, heads, causal): ViT-B/32 image tower, ViT-L/14, text tower (trimmed prompts), text tower full context, RN50 attention pool
for name, b, l, h, causal in [("vitb32", 512, 50, 12, 0), ("vitl14", 64, 257, 16, 0), ("text16", 10920, 16, 8, 1), ("text77", 2048, 77, 8, 1), ("rn50pool", 512, 50, 32, 0)]:
    w = h * 64
    qkv = (torch.randn(b * l, 3 * w, device="cuda") * 0.5).half()
    outs = [torch.zeros(b * l, w, dtype=torch.float16, device="cuda") for _ in libs]
    fns = [(lambda lib=lib, out=out: lib.hgr_mha(qkv.data_ptr(), out.data_ptr(), b, l, h, causal, 1, st)) for lib, out in zip(libs, outs)]
    for f in fns: assert f() == 0
    best = [1e9, 1e9]
    for rep in range(5):
        for i, f in enumerate(fns): best[i] = min(best[i], timeit(f))
    byts = b * l * w * 2 * 4
    print(json.dumps({"name": name, "old_us": round(best[0], 1), "new_us": round(best[1], 1), "new_gbps": round(byts / best[1] / 1e3), "equal": bool(torch.equal(outs[0], outs[1]))}))
