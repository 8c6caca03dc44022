#!/usr/bin/env python3
"""Dev tool: the image tower of the zero-shot step (ViT-B/32, batch 512) captured as HIP graphs under several settings of a
process-wide knob, replayed interleaved in ONE process (CDNA guide rule 24): what a kernel change is worth INSIDE the step.

    step_knob_ab.py tail           tail plan of gemm_nt_duo off / on (hgr_gemm_set_tail)
    step_knob_ab.py env NAME v1 v2 ...   an environment knob the host code re-reads per call
    step_knob_ab.py attr MODULE NAME v1 v2 ...   a module-level switch read at call time (e.g. attr hgr_net_amd.clip.model CLS_LAST 0 1)
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


what = sys.argv[1] if len(sys.argv) > 1 else "tail"
from hgr_net_amd import ops
if what == "tail":
    arms = [("tail_off", lambda: ops.gemm_set_tail(False)), ("tail_on", lambda: ops.gemm_set_tail(True, -1))]
elif what == "p8":           # gemm_nt_p8 by shape (2, the default: none of ViT-B/32's launches) / wherever it covers (1: c_fc, kv)
    from hgr_net_amd import _lib
    arms = [(f"p8={v}", (lambda v=v: _lib.load().hgr_gemm_set_p8(v))) for v in (2, 1)]
elif what == "attr":
    import importlib
    mod, name, vals = importlib.import_module(sys.argv[2]), sys.argv[3], sys.argv[4:]
    arms = [(f"{name}={v}", (lambda v=v: setattr(mod, name, type(getattr(mod, name))(int(v))))) for v in vals]
else:
    name, vals = sys.argv[2], sys.argv[3:]
    arms = [(f"{name}={v}", (lambda v=v: os.environ.__setitem__(name, v))) for v in vals]
graphs = {}
feats = {}
for label, setup in arms:
    setup()
    for _ in range(2):
        f = clip.encode_image(base)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f = clip.encode_image(base)
    graphs[label] = g
    g.replay()
    torch.cuda.synchronize()
    feats[label] = f.clone()
first = arms[0][0]
equal = {k: bool(torch.equal(v, feats[first])) for k, v in feats.items()}
maxdiff = {k: float((v - feats[first]).abs().max()) for k, v in feats.items()}
ts = {k: [] for k in graphs}
for rep in range(9):
    for k, g in graphs.items():
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        ts[k].append(s.elapsed_time(e) / 20)
print(json.dumps({"tower_ms_min_med": {k: [round(min(v), 3), round(sorted(v)[len(v) // 2], 3)] for k, v in ts.items()},
                  "features_equal_to_first_arm": equal, "max_abs_diff": maxdiff}), flush=True)
