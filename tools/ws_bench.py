#!/usr/bin/env python3
"""Dev tool (round 5): the role-split GEMM (hgr_gemm_nt_ws_exp, csrc/hgr_gemm_ws.hip) against gemm_nt_duo on the tower shapes.
Correctness first (exact equality with hgr_gemm_nt on small / multi-tile shapes), then interleaved timing rounds in one process."""
import sys, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops, _lib
from hgr_net_amd._lib import EPI_NONE

lib = C.CDLL(str(_lib.LIB_PATH))
fn = lib.hgr_gemm_nt_ws_exp
fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
fn.restype = C.c_int
lib.hgr_last_error.restype = C.c_char_p


def ws(a, w, out):
    m, k = a.shape
    n = w.shape[0]
    rc = fn(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), m, n, k,
            1 if a.dtype == torch.float16 else 0, torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError(lib.hgr_last_error().decode())
    return out


def mk(m, n, k, dt=torch.float16, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = ((torch.rand(m, k, device="cuda", generator=g) * 2 - 1)).to(dt)
    w = ((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * 0.05).to(dt)
    return a, w


def check(m, n, k, dt=torch.float16):
    a, w = mk(m, n, k, dt)
    ref = torch.empty(m, n, dtype=dt, device="cuda")
    got = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
    ops.gemm_nt(a, w, ref, epilogue=EPI_NONE)
    ws(a, w, got)
    torch.cuda.synchronize()
    f32 = (a.float() @ w.float().t())
    err_ref = (ref.float() - f32).abs().max().item()
    err_got = (got.float() - f32).abs().max().item()
    nbad = int((got != ref).sum().item()) if not torch.isnan(got).any() else -1
    print(json.dumps({"check": [m, n, k, str(dt)], "max_err_duo": err_ref, "max_err_ws": err_got, "n_differ": nbad,
                      "nan": bool(torch.isnan(got).any().item())}), flush=True)
    return nbad == 0


def timeit(f, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def bench(name, m, n, k, rounds=6, iters=20):
    a, w = mk(m, n, k)
    o1 = torch.empty(m, n, dtype=a.dtype, device="cuda")
    o2 = torch.empty(m, n, dtype=a.dtype, device="cuda")
    f1 = lambda: ops.gemm_nt(a, w, o1, epilogue=EPI_NONE)
    f2 = lambda: ws(a, w, o2)
    for _ in range(3):
        f1(); f2()
    torch.cuda.synchronize()
    t1, t2 = [], []
    for _ in range(rounds):
        t1.append(timeit(f1, iters))
        t2.append(timeit(f2, iters))
    fl = 2.0 * m * n * k
    r = {"shape": name, "duo_us": [round(min(t1), 1), round(sorted(t1)[len(t1) // 2], 1)], "ws_us": [round(min(t2), 1), round(sorted(t2)[len(t2) // 2], 1)],
         "duo_tf": round(fl / min(t1) / 1e6), "ws_tf": round(fl / min(t2) / 1e6), "equal": bool((o1 == o2).all().item())}
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    ok = True
    for shp in [(2048, 512, 256), (4096, 4096, 256), (2048, 1024, 768), (25600, 768, 768)]:
        ok = check(*shp) and ok
    ok = check(4096, 2048, 512, torch.bfloat16) and ok
    if not ok and "--force" not in sys.argv:
        print("MISMATCH - not timing"); sys.exit(1)
    for name, m, n, k in [("fc", 25600, 3072, 768), ("qkv", 25600, 2304, 768), ("out", 25600, 768, 768), ("proj", 25600, 768, 3072), ("sq4096", 4096, 4096, 4096)]:
        bench(name, m, n, k)
