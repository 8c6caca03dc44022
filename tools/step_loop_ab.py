#!/usr/bin/env python3
"""Dev tool: the WHOLE zero-shot evaluation step (Evaluator.add_images: image tower -> L2 norm -> hgr_logits_eval -> counters; ViT-B/32,
N = 21 841, batch 512 - bench.py's timed loop) under several settings of a switch, arms interleaved in ONE process on one box.

    step_loop_ab.py attr MODULE NAME v1 v2 ...     a module-level switch read at call time (e.g. attr hgr_net_amd.model.clip_tree TAIL_OVERLAP 0 1)
    step_loop_ab.py env NAME v1 v2 ...             an environment knob the host code re-reads per call

Prints one JSON line: ms per step (min / median over the rounds) per arm and whether the counters of every arm are identical.
"""
import json
import os
import sys
import tempfile
import time
import types
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from hgr_net_amd import evaluate, synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.hierarchy import build_hierarchy
from hgr_net_amd.model import tree_model

arch, nodes, batch = os.environ.get("AB_ARCH", "ViT-B/32"), int(os.environ.get("AB_NODES", "21841")), int(os.environ.get("AB_BATCH", "512"))
steps, rounds = int(os.environ.get("AB_STEPS", "30")), int(os.environ.get("AB_ROUNDS", "7"))
cfg = synth.CLIP_CONFIGS[arch]
edges = synth.make_dag(nodes, depth=12, seed=7, multi_parent=0.03)
h = build_hierarchy(edges)
n_test = int(round(nodes * 13442 / 20842))
splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], nodes - n_test, n_test, 13)
tokens = synth.make_tokens(nodes, 11, cfg["vocab_size"], n_ctx=0)
tmp = tempfile.mkdtemp(prefix="hgr_ab_")
gp = os.path.join(tmp, "graph.json")
json.dump(edges, open(gp, "w"))
opts = types.SimpleNamespace(device="cuda", folder=tmp, exp_name="HGR", weights="equal", out_ratio=0.25, in_ratio=0.5, from_epoch=-1,
                             graph_path=gp, arch=arch, fetch=False, load=False, load_path="none", scale=1.0, num_compare=256, k=1,
                             sample_strategy="topk", weighting="both", train_dtype="bf16", n_ctx=0)
clip = build_model(synth.clip_state_dict(cfg, 0)).to("cuda")
model = tree_model(opts, splits["all"], splits["rest"], node_tokens=tokens, clip_model=clip)
model.update_classifier()
base = synth.images(batch, cfg["image_resolution"], 1234).to("cuda")
bufs = [base, base.flip(0).contiguous()]
te = model.test_index.cpu().tolist()
targets = [te[(7 * i) % len(te)] for i in range(steps)]

what = sys.argv[1]
if what == "attr":
    import importlib
    mod, name, vals = importlib.import_module(sys.argv[2]), sys.argv[3], sys.argv[4:]
    arms = [(f"{name}={v}", (lambda v=v: setattr(mod, name, type(getattr(mod, name))(int(v))))) for v in vals]
else:
    name, vals = sys.argv[2], sys.argv[3:]
    arms = [(f"{name}={v}", (lambda v=v: os.environ.__setitem__(name, v))) for v in vals]


def loop(ev):
    for i in range(steps):
        ev.add_images(bufs[i & 1], targets[i])


counters, ts = {}, {k: [] for k, _ in arms}
for label, setup in arms:                       # warm-up + the counters of one pass per arm
    setup()
    ev = evaluate.Evaluator(model)
    loop(ev)
    counters[label] = ev.counters()
    torch.cuda.synchronize()
for r in range(rounds):
    for label, setup in arms:
        setup()
        ev = evaluate.Evaluator(model)
        loop(ev)                                # the arm's graphs are current again
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(ev)
        torch.cuda.synchronize()
        ts[label].append((time.perf_counter() - t0) / steps * 1e3)
first = arms[0][0]
print(json.dumps({"step_ms_min_med": {k: [round(min(v), 3), round(sorted(v)[len(v) // 2], 3)] for k, v in ts.items()},
                  "counters_equal_to_first_arm": {k: v == counters[first] for k, v in counters.items()}}), flush=True)
