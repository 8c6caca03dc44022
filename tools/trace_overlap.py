#!/usr/bin/env python3
"""Dev tool: does the class-token tail of step i really run beside the head of step i + 1?  Reads a rocprofv3 --kernel-trace CSV of
`tools/step_loop_ab.py attr hgr_net_amd.model.clip_tree TAIL_OVERLAP 1` and reports, per queue / stream, the busy intervals and how much
of the tail kernels' time (logits_eval / vit_head / eval_counters ...) overlaps kernels of another queue."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
cols = rows[0].keys()
qk = "Queue_Id" if "Queue_Id" in cols else None
sk = "Stream_Id" if "Stream_Id" in cols else None
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get(qk, "?") if qk else "?", r.get(sk, "?") if sk else "?"))
ev.sort()
t0 = ev[0][0]
byq = defaultdict(list)
for s, e, n, q, st in ev:
    byq[(q, st)].append((s, e, n))
print("columns:", list(cols))
for k, v in byq.items():
    print("queue/stream", k, "kernels", len(v), "busy ms", sum(e - s for s, e, _ in v) / 1e6, "span ms", (v[-1][1] - v[0][0]) / 1e6)
# overlap of every kernel whose name marks the tail with kernels on other queues
tail_marks = ("logits_eval_rows", "vit_head", "eval_counters", "l2norm")
others = sorted((s, e, q, st) for s, e, n, q, st in ev)
tot = ov = 0
for s, e, n, q, st in ev:
    if not any(m in n for m in tail_marks):
        continue
    tot += e - s
    for s2, e2, q2, st2 in others:
        if (q2, st2) == (q, st) or e2 <= s or s2 >= e:
            continue
        ov += min(e, e2) - max(s, s2)
print(f"tail kernels: {tot / 1e3:.1f} us total, {ov / 1e3:.1f} us of (kernel x concurrent kernel of another queue) overlap")
# the last 120 kernels as a timeline
for s, e, n, q, st in ev[-120:]:
    print(f"{(s - t0) / 1e3:12.1f} {(e - s) / 1e3:8.1f} us  q={q} st={st}  {n}")
