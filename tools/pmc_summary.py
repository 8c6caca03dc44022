#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of `bench.py` into profiles/: per-kernel time (kernel-trace --stats) and HBM
traffic per launch from the FETCH_SIZE / WRITE_SIZE PMC passes (collected in separate runs; both are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read stream - MI355X_MICROARCH.md "HBM" - so it is
doubled; the LayerNorm row calibrates that correction: 4 B read + 2 B written per element).

The summary records `kernel_source_hash` (sha256 over hgr-net_amd/csrc/*): bench.py quotes a summary's traffic only
when that hash equals the sources of the library it is running, so a stale profile can never be quoted silently.

usage: pmc_summary.py <tag> <stats_csv> <fetch_counter_csv> <write_counter_csv> [config]
       config = "" (ViT-B/32 zero-shot step, the headline), "rn50" (configs[2]) or "train_l14" (configs[4]): names the output
       profiles/<tag>_<config>_pmc_summary.json and selects which kernels count as the dominant family.
"""
import collections, csv, hashlib, json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted((ROOT / "hgr-net_amd" / "csrc").glob("*")):
        if f.suffix in (".hip", ".h") or f.name == "Makefile":
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    config = sys.argv[5] if len(sys.argv) > 5 else ""

    def short(n):
        n = n.replace("void ", "").replace("(anonymous namespace)::", "")
        return n.split("(")[0]

    def counters(path):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            d[short(r["Kernel_Name"]).replace("hgr_gemm::", "")].append(float(r["Counter_Value"]))
        return d

    f, w = counters(fetch), counters(write)
    rows = []
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        k = k.replace("hgr_gemm::", "")
        if not k.startswith(("gemm_nt", "gemm_tn", "layernorm", "mha_", "im2col", "topk", "level_argmax", "eval_rows", "logits_eval", "vit_embed", "l2norm", "text_embed",
                             "conv3x3", "stem_", "avgpool", "attnpool", "transpose", "colsum", "quickgelu", "adamw", "sumsq", "qkv_attn", "logits_slab", "vit_head",
                             "eval_counters", "rows_gather")):
            continue
        fv, wv = f.get(k, []), w.get(k, [])
        fetch_b = 2 * 1024 * sum(fv) / len(fv) if fv else None
        write_b = 1024 * sum(wv) / len(wv) if wv else None
        rows.append(dict(kernel=k, calls=int(r["Calls"]), avg_us=round(float(r["AverageNs"]) / 1e3, 1), pct=float(r["Percentage"]),
                         hbm_read_bytes_per_launch=round(fetch_b) if fetch_b else None,
                         hbm_write_bytes_per_launch=round(write_b) if write_b else None))
    # the tower GEMM family: gemm_nt_* / gemm_tn_* and (round 4) the in_proj GEMM fused with its attention, qkv_attn
    gem = [r for r in rows if r["kernel"].startswith(("gemm_nt", "gemm_tn", "qkv_attn")) and r["avg_us"] > 60 and r["hbm_read_bytes_per_launch"] and r["hbm_write_bytes_per_launch"] is not None]
    tot_calls = sum(r["calls"] for r in gem)
    out = dict(tag=tag, config=config or "vitb32", kernel_source_hash=kernel_source_hash(), note="FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as is; KiB -> bytes; averages per launch",
               kernels=rows,
               tower_gemm=dict(avg_us=round(sum(r["avg_us"] * r["calls"] for r in gem) / tot_calls, 1),
                               hbm_bytes_per_launch=round(sum((r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]) * r["calls"] for r in gem) / tot_calls)))
    json.dump(out, open(ROOT / "profiles" / (f"{tag}_{config}_pmc_summary.json" if config else f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out["tower_gemm"]))
    for r in rows[:10]:
        print(r)


if __name__ == "__main__":
    main()
