#!/usr/bin/env python3
"""Dev tool: per-launch time of one ModifiedResNet image-tower forward (eager, batch 512), grouped by op and shape.
usage: rn_layers.py [arch=RN50] [batch=512]"""
import collections
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import importlib  # noqa: E402

pkg = importlib.import_module("hgr_net_amd")
from hgr_net_amd import ops, synth  # noqa: E402
from hgr_net_amd.clip.model import build_model  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "RN50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = synth.CLIP_CONFIGS[arch]
model = build_model(synth.clip_state_dict(cfg, 0)).cuda()
img = torch.randn(batch, 3, cfg["image_resolution"], cfg["image_resolution"], device="cuda")
rec = []


def wrap(name, shape_of):
    fn = getattr(ops, name)

    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        rec.append((name, shape_of(*a, **k), e0, e1))
        return r
    setattr(ops, name, inner)


wrap("gemm_nt", lambda a, w, out, **k: (a.shape[0], w.shape[0], w.shape[1], k.get("epilogue", 0), 2.0 * a.shape[0] * w.shape[0] * w.shape[1]))
wrap("conv3x3_nhwc", lambda x, w, bias, out, b, h, wd, c, stride=1: (out.shape[0], w.shape[0], 9 * c, f"s{stride}", 2.0 * out.shape[0] * w.shape[0] * 9 * c))
for nm in ("stem_conv1", "stem_im2col", "avgpool2_nhwc", "attnpool_tokens", "attnpool_attend"):
    wrap(nm, lambda *a, **k: (tuple(a[-1].shape) if torch.is_tensor(a[-1]) else (tuple(a[1].shape) if torch.is_tensor(a[1]) else ()), 0.0))

for _ in range(3):
    rec.clear()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    model.encode_image(img)
    e.record()
    torch.cuda.synchronize()
agg = collections.OrderedDict()
tot = 0.0
for name, shp, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    key = (name,) + tuple(shp[:-1])
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += us
    d[2] += shp[-1]
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for key, (n, us, fl) in rows:
    print(json.dumps({"op": key[0], "shape": [str(x) for x in key[1:]], "calls": n, "us_total": round(us, 1), "us_each": round(us / n, 1),
                      "tflops": round(fl / us / 1e6, 1) if fl else None, "share": round(us / tot, 3)}))
print(json.dumps({"arch": arch, "batch": batch, "sum_of_launches_us": round(tot, 1), "wall_us": round(s.elapsed_time(e) * 1e3, 1),
                  "img_per_s_tower": round(batch / (s.elapsed_time(e) * 1e-3))}))
