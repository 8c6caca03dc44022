#!/usr/bin/env python3
"""Dev tool: the image tower (ViT-B/32, batch 512 by default) as 1, 2, 3, 4 independent batch slices on as many streams, each
variant captured as a HIP graph, replayed interleaved in one process.  Checks that the features are bit-identical.

    img_streams_ab.py [--arch ViT-B/32] [--batch 512] [streams ...]
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import synth
from hgr_net_amd.clip import model as cm

args = sys.argv[1:]
arch = args[args.index("--arch") + 1] if "--arch" in args else "ViT-B/32"
batch = int(args[args.index("--batch") + 1]) if "--batch" in args else 512
skip = set()
for f in ("--arch", "--batch"):
    if f in args:
        skip |= {args.index(f), args.index(f) + 1}
variants = [int(x) for i, x in enumerate(args) if i not in skip] or [1, 2, 3, 4]

cfg = synth.CLIP_CONFIGS[arch]
clip = cm.build_model(synth.clip_state_dict(cfg, 0)).to("cuda")
img = synth.images(batch, cfg["image_resolution"], 1).to("cuda")
graphs, outs = {}, {}
for ns in variants:
    cm.IMG_STREAMS = ns
    for _ in range(2):
        o = clip.encode_image(img)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o = clip.encode_image(img)
    g.replay()
    torch.cuda.synchronize()
    graphs[ns], outs[ns] = g, o.clone()
ref = outs[variants[0]]
res = {"arch": arch, "batch": batch, "equal": {ns: bool(torch.equal(outs[ns], ref)) for ns in variants}}
ts = {ns: [] for ns in variants}
for rep in range(7):
    for ns in variants:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            graphs[ns].replay()
        e.record()
        torch.cuda.synchronize()
        ts[ns].append(s.elapsed_time(e) / 10)
for ns in variants:
    t = sorted(ts[ns])
    res[f"ms_{ns}"] = [round(t[0], 3), round(t[len(t) // 2], 3)]
print(json.dumps(res), flush=True)
