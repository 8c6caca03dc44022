#!/usr/bin/env python3
"""Dev tool: where one zero-shot step of bench.py goes, in one process, interleaved rounds (ViT-B/32, N = 21 841, batch 512):
   tower      - clip_model.encode_image replayed as a HIP graph
   fwd_eval   - tree_model.forward_eval (tower + L2 norm + hgr_logits_eval) replayed as a HIP graph
   step       - Evaluator.add_images (fwd_eval + the counters kernel launched eagerly behind it), as bench.py times it
   step_2buf  - the same alternating between two input buffers (two graphs), exactly bench.py's loop
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
model.update_classifier()
ev = evaluate.Evaluator(model)
base = synth.images(batch, cfg["image_resolution"], 1234).to("cuda")
bufs = [base, base.flip(0).contiguous()]
te = model.test_index.cpu().tolist()

for _ in range(2):
    clip.encode_image(base)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    clip.encode_image(base)
plan = None


def tower():
    g.replay()


def fwd_eval():
    model.forward_eval(base, ev._plan, 20)


def step():
    ev.add_images(base, te[3])


cnt = [0]


def step_2buf():
    cnt[0] += 1
    ev.add_images(bufs[cnt[0] & 1], te[cnt[0] % len(te)])


for _ in range(3):
    step(); step_2buf()
fns = {"tower": tower, "fwd_eval": fwd_eval, "step": step, "step_2buf": step_2buf}
ts = {k: [] for k in fns}
for rep in range(7):
    for k, f in fns.items():
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record()
        torch.cuda.synchronize()
        ts[k].append(s.elapsed_time(e) / 20)
print(json.dumps({k: [round(min(v), 3), round(sorted(v)[len(v) // 2], 3)] for k, v in ts.items()}), flush=True)
