#!/usr/bin/env python3
"""Dev tool: time the device transform (hgr_preprocess_bicubic) on an ImageNet-like batch and the same work in Pillow on
the host cores.  usage: preproc_bench.py [B]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from hgr_net_amd import preprocess

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(0)
shapes = [(375, 500), (500, 375), (333, 500), (500, 400), (480, 640), (600, 800)]
imgs = [rng.integers(0, 256, shapes[i % len(shapes)] + (3,), dtype=np.uint8) for i in range(B)]
src_bytes = sum(im.size for im in imgs)
pre = preprocess.BatchPreprocessor(224, "cuda")
for out in ("u8", "f32"):
    pre(imgs, output=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pre(imgs, output=out)
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / 5
    # kernel alone: tables already resident
    from hgr_net_amd import ops
    prof = []
    import hgr_net_amd.ops as o
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    orig = o.preprocess_bicubic
    def timed(*a, **k):
        ev0.record(); orig(*a, **k); ev1.record()
    o.preprocess_bicubic = timed
    preprocess.ops.preprocess_bicubic = timed
    pre(imgs, output=out); torch.cuda.synchronize()
    kern = ev0.elapsed_time(ev1) * 1e-3
    o.preprocess_bicubic = orig; preprocess.ops.preprocess_bicubic = orig
    out_bytes = B * 224 * 224 * 3 * (1 if out == "u8" else 4)
    print(json.dumps({"output": out, "B": B, "host+h2d+kernel_ms": round(whole * 1e3, 2), "kernel_us": round(kern * 1e6, 1),
                      "img_per_s_kernel": round(B / kern), "img_per_s_whole": round(B / whole),
                      "kernel_GBps": round((src_bytes + out_bytes) / kern / 1e9, 1), "src_MB": round(src_bytes / 1e6, 1)}))
# host baseline: Pillow resize + crop + normalise, one thread
from PIL import Image
t0 = time.perf_counter()
n = min(B, 64)
for im in imgs[:n]:
    pil = Image.fromarray(im)
    w, h = pil.size
    nw, nh = preprocess.resized_size(w, h, 224)
    pil = pil.resize((nw, nh), Image.BICUBIC)
    l, t = preprocess.crop_origin(nw, nh, 224)
    a = np.asarray(pil.crop((l, t, l + 224, t + 224)), dtype=np.float32) / 255.0
    x = (torch.from_numpy(a).permute(2, 0, 1) - torch.tensor(preprocess.ops.CLIP_MEAN).view(3, 1, 1)) / torch.tensor(preprocess.ops.CLIP_STD).view(3, 1, 1)
cpu = (time.perf_counter() - t0) / n
print(json.dumps({"pillow_1thread_img_per_s": round(1 / cpu, 1)}))
