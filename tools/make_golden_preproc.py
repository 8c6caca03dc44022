#!/usr/bin/env python3
"""Generates tests/golden/preproc.npz: inputs and expected outputs of the reference's image transform
(clip/clip.py:71-78 = dataset/imagenet_group.py:27-34), produced by the real third-party code the reference calls:
Pillow's `Image.resize(BICUBIC)` / `crop` (installed here) and torch's ToTensor / Normalize arithmetic.  torchvision is
not installed in this image; its `Resize(int)` and `CenterCrop` size rules are the two one-liners below
(torchvision/transforms/functional_pil.py `resize`, functional.py `center_crop`).  Run in the build container only."""
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "preproc.npz"
CASES = [(37, 53, 16), (53, 37, 16), (100, 75, 32), (64, 48, 32), (33, 32, 32), (32, 200, 32), (17, 17, 32), (150, 100, 64),
         (64, 64, 64), (90, 64, 64)]          # (H, W, n_px): down- and up-scaling, both orientations, no-op sides
MEAN = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)
STD = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)


def reference_transform(arr: np.ndarray, n_px: int):
    img = Image.fromarray(arr).convert("RGB")
    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    if short != n_px:                                                   # Resize(n_px, BICUBIC)
        new_short, new_long = n_px, int(n_px * long / short)
        img = img.resize((new_short, new_long) if w <= h else (new_long, new_short), Image.BICUBIC)
    w, h = img.size                                                     # CenterCrop(n_px)
    top, left = int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))
    img = img.crop((left, top, left + n_px, top + n_px))
    u8 = np.asarray(img)
    t = torch.from_numpy(u8.copy()).permute(2, 0, 1).contiguous().float().div(255)   # ToTensor
    t = (t - MEAN) / STD                                                                # Normalize
    return u8, t.numpy()


def main():
    rng = np.random.default_rng(20260103)
    out = {}
    for i, (h, w, n) in enumerate(CASES):
        # smooth gradient + noise + saturated patches: exercises the clip at 0 / 255 of the bicubic overshoot
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 7 % 256)], -1)
        img = np.clip(base + rng.integers(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
        img[: h // 4, : w // 4] = 255
        img[-(h // 4):, -(w // 4):] = 0
        u8, f32 = reference_transform(img, n)
        out[f"in_{i}"] = img
        out[f"u8_{i}"] = u8
        out[f"f32_{i}"] = f32
        out[f"npx_{i}"] = np.int32(n)
    out["n_cases"] = np.int32(len(CASES))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    sys.exit(main())
