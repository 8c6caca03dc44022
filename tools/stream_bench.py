#!/usr/bin/env python3
"""Dev tool: time hgr_gemm_nt on the class-logits shape (512 x 21841 x 512, fp32 out) in this process; run once per
HGR_GEMM_STREAM setting (read once per process).  usage: stream_bench.py [M N K [pad]]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hgr_net_amd import ops
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 21841, 512)
pad = int(sys.argv[4]) if len(sys.argv) >= 5 else 0          # extra elements per operand row (row pitch = 2 * (K + pad) bytes)
a = (torch.rand(m, k + pad, device="cuda") * 2 - 1).half()[:, :k]
w = ((torch.rand(n, k + pad, device="cuda") * 2 - 1) * 0.05).half()[:, :k]
out = torch.zeros(m, (n + 63) // 64 * 64, dtype=torch.float32, device="cuda")
for _ in range(3):
    ops.gemm_nt(a, w, out, n=n)
best = 1e9
for rep in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.gemm_nt(a, w, out, n=n)
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 20 * 1e3)
print(json.dumps({"stream": os.environ.get("HGR_GEMM_STREAM", "-"), "lib": os.environ.get("HGR_LIB", "in-tree"), "shape": [m, n, k], "pad": pad,
                  "us": round(best, 2), "tflops": round(2.0 * m * n * k / best / 1e6), "gbps": round((2 * m * k + 2 * n * k + 4 * m * n) / best / 1e3)}))
