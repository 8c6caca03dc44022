#!/usr/bin/env python3
"""Dev tool: per-parameter gradient norms of one OM step on the tiny-rn fixture vs the reference's."""
import sys, json, types, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from hgr_net_amd import synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.hierarchy import build_hierarchy
from hgr_net_amd.model import tree_model
G = Path(__file__).resolve().parent.parent / "tests" / "golden"
case = sys.argv[1] if len(sys.argv) > 1 else "tinyrn_n64"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
meta = json.load(open(G / f"tree_{case}.json")); z = np.load(G / f"tree_{case}.npz"); gold = np.load(G / f"train_{case}.npz")
cfg, d, t = meta["config"], meta["dag"], meta["train"]
edges = synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"])
tmp = Path(tempfile.mkdtemp()); (tmp / "g.json").write_text(json.dumps(edges))
h = build_hierarchy(edges)
splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
o = types.SimpleNamespace(device="cuda", folder=str(tmp / "out"), exp_name="HGR", weights="equal", from_epoch=-1, graph_path=str(tmp / "g.json"),
                          arch="synthetic", fetch=False, load=False, load_path="none", scale=1.0, train_dtype=dtype, **t["opts"])
model = tree_model(o, splits["all"], splits["rest"], node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)),
                   clip_model=build_model(synth.clip_state_dict(cfg, 0)).to("cuda"))
img = synth.images(t["bsz"], cfg["image_resolution"], t["image_seed"]).to("cuda")
targets = torch.full((t["bsz"],), t["target"], dtype=torch.long, device="cuda")
model.train_batch(img, targets, "OM", "topk")
for p in model.parameters(): p.grad = None
model._trainer.contra_override = lambda i: tuple(t["contra"][i])
loss = model.train_batch(img, targets, "OM", "topk")
print("loss", loss, "ref", t["loss"])
named = dict(model.clip_model.named_parameters())
for k, ref in t["grad_norms"].items():
    g = named[k].grad
    got = float(g.norm()) if g is not None else float("nan")
    flag = "" if abs(got - ref) <= 0.08 * ref + 1e-4 else "   <<<<"
    cos = ""
    if "grad/" + k in gold.files and g is not None:
        a, b = torch.from_numpy(gold["grad/" + k]).flatten(), g.detach().cpu().flatten()
        cos = " cos %.4f" % float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
    if k.startswith("visual") or flag:
        print(f"{k:55s} got {got:.5e} ref {ref:.5e}{cos}{flag}")
