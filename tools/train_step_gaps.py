#!/usr/bin/env python3
"""Dev tool: one ViT-L/14 + CoOp OM training step of bench.py under the torch profiler - GPU busy / idle time inside the step, the largest
gaps with the kernel that follows each (host synchronisations show up as 0.1 - 1 ms gaps before a copy), and the per-kernel totals of
the last step.  Run on the GPU box: `python tools/train_step_gaps.py`."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import runpy

import torch
from torch.profiler import ProfilerActivity, profile

os.chdir(ROOT)
sys.argv = ["bench.py", "--mode", "train", "--arch", "ViT-L/14", "--n-ctx", "16", "--train-dtype", "bf16", "--batch", "256", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-pcie", "--no-c1", "--secondary"]
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    try:
        runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
    except SystemExit:
        pass
ev = [(e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
ev.sort()
print("device events", len(ev))
# the last adamw marks the end of the last step; the one before it the end of the previous step
ad = [i for i, e in enumerate(ev) if "adamw" in e[2]]
print("adamw idx", ad)
lo, hi = ad[-2] + 1, ad[-1]
t0, t1 = ev[lo][0], ev[hi][1]
busy = 0.0; cur_end = t0; gaps = []
for s, e, n in ev[lo:hi + 1]:
    if s > cur_end:
        gaps.append((s - cur_end, n))
    busy += max(0.0, e - max(s, cur_end)); cur_end = max(cur_end, e)
print(f"last step window {(t1 - t0) / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {(t1 - t0 - busy) / 1e3:.2f} ms, kernels {hi - lo + 1}")
gaps.sort(reverse=True)
print("largest gaps (us, next kernel):")
for g, n in gaps[:25]: print(f"  {g:9.1f}  {n[:100]}")
import collections
hist = collections.Counter()
for g, n in gaps: hist[min(int(g // 5) * 5, 100)] += g
print("idle by gap size bucket (us -> total ms):", {k: round(v / 1e3, 2) for k, v in sorted(hist.items())})
by = collections.defaultdict(float)
for s, e, n in ev[lo:hi + 1]: by[n[:70]] += e - s
for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:30]: print(f"  {v / 1e3:8.2f} ms  {n}")
