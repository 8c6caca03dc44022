#!/usr/bin/env python3
"""Dev tool: idle time between the kernels of one evaluation step, from a `rocprofv3 --kernel-trace` CSV of bench.py:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-secondary --no-c1
    python tools/trace_gaps.py gpurun_out/trace/.../t_kernel_trace.csv

Steps are found as the repeating runs that start with the patch unfold (im2col_vec8) and contain 11 full attention launches; per step:
wall time first start -> last end, the sum of kernel durations, the sum of the gaps and the largest ones with the kernels around them."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "im2col_vec8" in e[2]]
steps = []
for a, b in zip(starts, starts[1:]):
    seg = ev[a:b]
    if sum("mha_fwd" in e[2] for e in seg) in (11, 12) and len(seg) < 120:
        steps.append(seg)
print(f"{len(steps)} steps found, {len(steps[0]) if steps else 0} kernels each")
for seg in steps[-4:]:
    wall = (seg[-1][1] - seg[0][0]) / 1e3
    busy = sum(e[1] - e[0] for e in seg) / 1e3
    gaps = [((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"wall {wall:.1f} us, kernels {busy:.1f} us, gaps {sum(g[0] for g in pos):.1f} us over {len(pos)} boundaries (median {sorted(g[0] for g in pos)[len(pos) // 2]:.2f} us), overlap {-sum(g[0] for g in gaps if g[0] < 0):.1f} us")
for g in sorted(gaps, reverse=True)[:8]:
    print(f"  {g[0]:7.2f} us  {g[1]} -> {g[2]}")
