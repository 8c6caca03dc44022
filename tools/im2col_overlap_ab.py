#!/usr/bin/env python3
"""Dev tool: does the patch unfold (im2col, HBM-bound, 75 us) hide beside the previous batch's MFMA-bound tower kernels when it runs
on a side stream one batch ahead (CLIP.prestage_patches)?  ViT-B/32, batch 512, one process, interleaved rounds:
   serial   - encode_image as one HIP graph (unfold first, then the tower), back to back
   overlap  - the tower WITHOUT its unfold as a HIP graph on the main stream; the unfold of the next batch is launched on a side
              stream right after each replay; the next replay waits for it
"""
import json
import os
import sys
import tempfile
import types
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import evaluate, synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.hierarchy import build_hierarchy
from hgr_net_amd.model import tree_model

arch, nodes, batch = "ViT-B/32", 21841, 512
cfg = synth.CLIP_CONFIGS[arch]
edges = synth.make_dag(nodes, depth=12, seed=7, multi_parent=0.03)
h = build_hierarchy(edges)
n_test = int(round(nodes * 13442 / 20842))
splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], nodes - n_test, n_test, 13)
tokens = synth.make_tokens(nodes, 11, cfg["vocab_size"], n_ctx=0)
tmp = tempfile.mkdtemp(prefix="hgr_bd_")
gp = os.path.join(tmp, "graph.json")
json.dump(edges, open(gp, "w"))
opts = types.SimpleNamespace(device="cuda", folder=tmp, exp_name="HGR", weights="equal", out_ratio=0.25, in_ratio=0.5, from_epoch=-1,
                             graph_path=gp, arch=arch, fetch=False, load=False, load_path="none", scale=1.0, num_compare=256, k=1,
                             sample_strategy="topk", weighting="both", train_dtype="bf16", n_ctx=0)
clip = build_model(synth.clip_state_dict(cfg, 0)).to("cuda")
model = tree_model(opts, splits["all"], splits["rest"], node_tokens=tokens, clip_model=clip)
base = synth.images(batch, cfg["image_resolution"], 1234).to("cuda")
bufs = [base, base.flip(0).contiguous()]
te = model.test_index.cpu().tolist()


bufs = [base, base.flip(0).contiguous()]
side = torch.cuda.Stream()
for _ in range(2):
    clip.encode_image(base)
torch.cuda.synchronize()
g_full = []
for x in bufs:
    clip.encode_image(x)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f = clip.encode_image(x)
    g_full.append((g, f))
g_rest = []
for i, x in enumerate(bufs):
    clip.prestage_patches(x, slot=i)
    clip.encode_image(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f = clip.encode_image(x)
    g_rest.append((g, f))
clip.prestage_patches(None)
torch.cuda.synchronize()
g_full[0][0].replay(); g_rest[0][0].replay(); torch.cuda.synchronize()
print("features equal:", bool(torch.equal(g_full[0][1], g_rest[0][1])), flush=True)
main = torch.cuda.current_stream()


def serial(n):
    for i in range(n):
        g_full[i & 1][0].replay()


def overlap(n):
    ready = clip.prestage_patches(bufs[0], slot=0, stream=side)
    for i in range(n):
        main.wait_event(ready)
        g_rest[i & 1][0].replay()
        if i:                                                # slot (i + 1) & 1 was last read by batch i - 1's patch GEMM
            side.wait_event(done_prev)
        done_prev = torch.cuda.Event()
        done_prev.record(main)
        ready = clip.prestage_patches(bufs[(i + 1) & 1], slot=(i + 1) & 1, stream=side)
    clip.prestage_patches(None)


fns = {"serial": serial, "overlap": overlap}
ts = {k: [] for k in fns}
for rep in range(7):
    for k, f in fns.items():
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        f(20)
        e.record()
        torch.cuda.synchronize()
        ts[k].append(s.elapsed_time(e) / 20)
print(json.dumps({k: [round(min(v), 3), round(sorted(v)[len(v) // 2], 3)] for k, v in ts.items()}), flush=True)
