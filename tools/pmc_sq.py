#!/usr/bin/env python3
"""Dev tool: per-kernel averages of arbitrary rocprofv3 PMC counters (one pass: `rocprofv3 --kernel-trace --pmc A B C ...
--output-format csv -d DIR -- python3 <cmd>`), with the derived quantities the CDNA guide names:

    clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration             (sum over the 8 XCDs; reads high on short dispatches)
    mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
    wait_any    = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (quad-cycles both)

usage: pmc_sq.py DIR [name-prefix ...]"""
import collections, csv, json, sys
from pathlib import Path

d = Path(sys.argv[1])
prefixes = tuple(sys.argv[2:]) or ("gemm_nt", "mha_fwd", "logits_eval")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


cc = next(d.rglob("*counter_collection.csv"))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    k = short(r["Kernel_Name"])
    if not k.startswith(prefixes):
        continue
    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if "Start_Timestamp" in r and r.get("Dispatch_Id"):
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
kt = list(d.rglob("*kernel_trace.csv"))
if kt and not any(dur.values()):
    for r in csv.DictReader(open(kt[0])):
        k = short(r["Kernel_Name"])
        if k.startswith(prefixes):
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, c in sorted(vals.items(), key=lambda kv: -sum(dur[kv[0]].values())):
    n = max(len(v) for v in c.values())
    avg = {cn: sum(v) / len(v) for cn, v in c.items()}
    us = sum(dur[k].values()) / max(1, len(dur[k]))
    o = {"kernel": k, "launches": n, "avg_us": round(us, 1)}
    g = avg.get("GRBM_GUI_ACTIVE")
    if g and us:
        o["clock_ghz"] = round(g / 8 / us / 1e3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            o["mfma_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8 * 1024), 3)
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for cn, nm in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active_inst"), ("SQ_WAIT_INST_LDS", "wait_inst_lds")):
            if cn in avg:
                o[nm] = round(avg[cn] / wc, 3)
    o["raw"] = {cn: round(v) for cn, v in avg.items()}
    print(json.dumps(o))
