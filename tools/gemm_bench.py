#!/usr/bin/env python3
"""Dev tool (GPU box): time hgr_gemm_nt on the tower / logits shapes, check it against a matmul of the
same 16-bit inputs, and print torch.matmul's (hipBLASLt) time on the same data as a known-good ceiling.
Not part of the product or of bench.py."""
import sys, json, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import ops
from hgr_net_amd._lib import EPI_NONE, EPI_BIAS, EPI_BIAS_QUICKGELU, EPI_BIAS_RESIDUAL

SHAPES = [  # name, M, N, K, epilogue, out32
    ("qkv", 25600, 2304, 768, EPI_BIAS, False),
    ("out", 25600, 768, 768, EPI_BIAS_RESIDUAL, True),
    ("fc", 25600, 3072, 768, EPI_BIAS_QUICKGELU, False),
    ("proj", 25600, 768, 3072, EPI_BIAS_RESIDUAL, True),
    ("patch", 25088, 768, 3072, EPI_NONE, True),
    ("logits", 512, 21841, 512, EPI_NONE, True),
    ("sq4096", 4096, 4096, 4096, EPI_NONE, False),
]

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    dt = torch.float16 if "--bf16" not in sys.argv else torch.bfloat16
    dev = "cuda"
    rows = []
    for name, m, n, k, epi, out32 in SHAPES:
        torch.manual_seed(0)
        a = (torch.rand(m, k, device=dev) * 2 - 1).to(dt)
        w = ((torch.rand(n, k, device=dev) * 2 - 1) * 0.05).to(dt)
        bias = torch.rand(n, device=dev)
        res = torch.rand(m, n, device=dev) if epi == EPI_BIAS_RESIDUAL else None
        ld = (n + 63) // 64 * 64
        out = torch.empty(m, ld, dtype=torch.float32 if out32 else dt, device=dev)
        x = res.clone() if res is not None else None
        def run():
            if epi == EPI_BIAS_RESIDUAL:
                ops.gemm_nt(a, w, x, bias=bias, residual=x, epilogue=epi)
            else:
                ops.gemm_nt(a, w, out, bias=bias if epi != EPI_NONE else None, epilogue=epi, n=n)
        if epi == EPI_BIAS_RESIDUAL:
            x.copy_(res); run(); got = x.float()
            ref = res + a.float() @ w.float().t() + bias
        else:
            run(); got = out[:, :n].float()
            ref = a.float() @ w.float().t()
            if epi != EPI_NONE: ref = ref + bias
            if epi == EPI_BIAS_QUICKGELU: ref = ref * torch.sigmoid(1.702 * ref)
        err = float((got - ref).abs().max()); scale = float(ref.abs().max())
        t = timeit(run)
        tb = timeit(lambda: torch.matmul(a, w.t()))
        fl = 2.0 * m * n * k
        rows.append(dict(name=name, M=m, N=n, K=k, us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1),
                         blas_us=round(tb * 1e6, 1), blas_tflops=round(fl / tb / 1e12, 1), err=err, scale=scale))
        print(json.dumps(rows[-1]), flush=True)

if __name__ == "__main__":
    main()
