#!/usr/bin/env python3
"""Dev tool: wall time of one OM training step (ViT-B/32, batch 256, N = 21 841 hierarchy) on one GPU."""
import sys, json, time, tempfile, types, os, random
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.hierarchy import build_hierarchy
from hgr_net_amd.model import tree_model
from hgr_net_amd.training import FusedAdamW
arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NCTX = int(sys.argv[3]) if len(sys.argv) > 3 else 0
N = 21841
cfg = synth.CLIP_CONFIGS[arch]
edges = synth.make_dag(N, 12, 7, 0.03); h = build_hierarchy(edges)
splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], 7400, 13442, 13)
tmp = tempfile.mkdtemp(); gp = os.path.join(tmp, "g.json"); json.dump(edges, open(gp, "w"))
o = types.SimpleNamespace(device="cuda:0", folder=tmp, exp_name="HGR", weights="equal", out_ratio=0.25, in_ratio=0.5, from_epoch=-1, graph_path=gp,
                          arch=arch, fetch=False, load=False, load_path="none", scale=1.0, num_compare=256, k=1, sample_strategy="topk", weighting="both", n_ctx=NCTX)
model = tree_model(o, splits["all"], splits["rest"], node_tokens=synth.make_tokens(N, 11, n_ctx=NCTX), clip_model=build_model(synth.clip_state_dict(cfg, 0)).to("cuda:0"))
img = synth.images(B, 224, 5).to("cuda:0")
deep = max(model.train_index.tolist(), key=lambda i: len(model.c2p[i]))
tg = torch.full((B,), deep, dtype=torch.long, device="cuda:0")
params = [p for n, p in model.named_parameters() if p.requires_grad and n != "layer_weight"]
opt = FusedAdamW(params, lr=3e-7)
random.seed(0)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    opt.zero_grad(); loss = model.train_batch(img, tg, "OM", "topk"); opt.step()
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{arch} B={B} n_ctx={NCTX} step {it}: loss {loss:.4f}  {dt*1e3:.1f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB  inner steps {len(model._trainer.last_contra)}  depth {len(model.c2p[deep])}", flush=True)
