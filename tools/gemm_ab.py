#!/usr/bin/env python3
"""Dev tool: A/B two builds of libhgr's hgr_gemm_nt in ONE process, interleaved (box-to-box clock differences between
gpurun calls are larger than the effects being measured).  usage: gemm_ab.py old.so new.so [tile]"""
import sys, ctypes as C, json
import torch
_p, _l, _i = C.c_void_p, C.c_int64, C.c_int
SIG = [_p, _l, _p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p]
libs = []
for path in sys.argv[1:3]:
    lib = C.CDLL(path); lib.hgr_gemm_nt.argtypes = SIG; lib.hgr_gemm_nt.restype = _i; libs.append(lib)
SHAPES = [("qkv", 25600, 2304, 768, 1, 0), ("out", 25600, 768, 768, 3, 1), ("fc", 25600, 3072, 768, 2, 0), ("proj", 25600, 768, 3072, 3, 1),
          ("patch", 25088, 768, 3072, 0, 1), ("logits", 512, 21841, 512, 0, 1), ("sq4096", 4096, 4096, 4096, 0, 0), ("k8192", 4096, 4096, 8192, 0, 0)]
st = torch.cuda.current_stream().cuda_stream
def run(lib, a, w, out, bias, res, m, n, k, epi, o32):
    rc = lib.hgr_gemm_nt(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), bias.data_ptr() if epi else None,
                         res.data_ptr() if epi == 3 else None, out.stride(0) if epi == 3 else 0, m, n, k, 1, epi, o32, st)
    assert rc == 0
def timeit(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, m, n, k, epi, o32 in SHAPES:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).half(); w = ((torch.rand(n, k, device="cuda") * 2 - 1) * 0.05).half()
    bias = torch.rand(n, device="cuda"); ld = (n + 63) // 64 * 64
    outs = [torch.zeros(m, ld, dtype=torch.float32 if o32 else torch.float16, device="cuda") for _ in libs]
    fns = [(lambda lib=lib, out=out: run(lib, a, w, out, bias, out, m, n, k, epi, o32)) for lib, out in zip(libs, outs)]
    for f in fns: f(); f()
    best = [1e9, 1e9]
    for rep in range(5):
        for i, f in enumerate(fns): best[i] = min(best[i], timeit(f))
    same = None
    if epi != 3:
        same = bool(torch.equal(outs[0][:, :n], outs[1][:, :n]))
    print(json.dumps({"name": name, "old_us": round(best[0], 1), "new_us": round(best[1], 1), "new_tflops": round(2.0 * m * n * k / best[1] / 1e6), "equal": same}))
