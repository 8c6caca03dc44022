#!/usr/bin/env python3
"""Generates tests/golden/loader.json + loader_imgs.npz by RUNNING the reference's group loaders
(dataset/imagenet_group.py, dataset/imagenet_group_test.py) in this container: sampler orders for fixed seeds and the
batches `DataManager_test` yields over a tiny on-disk dataset (lossless PNG files written from the arrays stored in
the fixture).  torchvision is absent; its transform is replaced, on the reference object, by the same Pillow + torch
arithmetic tools/make_golden_preproc.py uses.  Nothing of the reference travels: the fixture holds inputs and outputs."""
import json
import os
import random
import sys
import tempfile
import types
from pathlib import Path

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("HGR_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))

import numpy as np
import torch
from PIL import Image

from make_golden import install_stubs
from make_golden_preproc import reference_transform

GOLD = REPO / "tests" / "golden"
N_PX = 16
CLASSES = {"n001": [(20, 30), (31, 17), (16, 16), (40, 25), (18, 50)], "n002": [], "n003": [(25, 25), (33, 21)],
           "n004": [(17, 29), (29, 17), (64, 48)]}        # class -> image sizes (H, W); n002 is empty on purpose


def main():
    install_stubs()
    sys.path.insert(0, str(REF))
    from dataset import imagenet_group as ref_train
    from dataset import imagenet_group_test as ref_test

    out = {"n_px": N_PX, "train_sampler": [], "classes": {}}
    # 1. training sampler order: global `random`, seeded
    for seed, n_episodes, n_groups in [(0, 7, 3), (5, 10, 4), (9, 3, 5)]:
        random.seed(seed)
        seq = [g[0] for g in ref_train.GroupBatchSampler(n_episodes, n_groups)]
        out["train_sampler"].append({"seed": seed, "n_episodes": n_episodes, "n_groups": n_groups, "order": seq,
                                     "len": len(ref_train.GroupBatchSampler(n_episodes, n_groups))})

    # 2. the evaluation loader over real files
    rng = np.random.default_rng(77)
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        split = {}
        for cls, sizes in CLASSES.items():
            split[cls] = []
            for j, (h, w) in enumerate(sizes):
                a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                name = f"{cls}_{j}.png"
                Image.fromarray(a).save(os.path.join(tmp, name))
                arrays[name] = a
                split[cls].append(os.path.join(tmp, name))
        os.makedirs(os.path.join(tmp, "data"))
        json.dump(split, open(os.path.join(tmp, "data", "val_split.json"), "w"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            node_set = ["n000", "n001", "n002", "n003", "n004", "n005"]
            opts = types.SimpleNamespace(test_batch_size=2)
            dm = ref_test.DataManager_test(opts=opts, split="val", node_set=node_set, candidates=["n004", "n001", "n002", "n003"],
                                           resolution=N_PX)
            dm.transform = lambda img: torch.from_numpy(reference_transform(np.asarray(img.convert("RGB")), N_PX)[1])
            # the reference hard-codes num_workers=12; the tiny fixture run uses in-process loading (same batches)
            orig = ref_test.DataLoader
            ref_test.DataLoader = lambda *a, **k: orig(*a, **{**k, "num_workers": 0, "pin_memory": False})
            loader = dm.get_data_loader()
            batches = []
            imgs = {}
            for i, data in enumerate(loader):
                paths = [os.path.basename(p[0]) for p in data["path"]]
                batches.append({"label": data["label"][0].tolist(), "paths": paths, "img_shape": list(data["img"].shape)})
                imgs[f"batch_{i}"] = data["img"].numpy()
            out["test"] = {"batch_size": 2, "node_set": node_set, "candidates": ["n004", "n001", "n002", "n003"],
                           "num_batch": loader.batch_sampler.num_batch, "num_data": dm.num_data, "batches": batches}
            ref_test.DataLoader = orig
        finally:
            os.chdir(cwd)
    out["classes"] = {cls: [f"{cls}_{j}.png" for j in range(len(s))] for cls, s in CLASSES.items()}
    json.dump(out, open(GOLD / "loader.json", "w"), indent=1)
    np.savez_compressed(GOLD / "loader_imgs.npz", **{"src_" + k: v for k, v in arrays.items()}, **imgs)
    print("batches:", [(b["label"], b["paths"]) for b in out["test"]["batches"]])
    print("wrote", GOLD / "loader.json")


if __name__ == "__main__":
    main()
