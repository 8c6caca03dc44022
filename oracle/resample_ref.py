"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's image transform (never imported by the product path).

The reference builds its input tensor with (clip/clip.py:71-78 and dataset/imagenet_group.py:27-34)

    Compose([Resize(n_px, interpolation=BICUBIC), CenterCrop(n_px), convert("RGB"), ToTensor(), Normalize(mean, std)])

on a PIL image that dataset/imagenet_group.py:156 has already converted to RGB.  The arithmetic lives in two third-party
dependencies that are NOT vendored in /root/reference:

  * torchvision (README.md pins pytorch=1.7.1 -> torchvision 0.8.x): `Resize(int)` on a PIL image resizes the SHORT side
    to n_px and the long side to int(n_px * long / short) (transforms/functional_pil.py `resize`); `CenterCrop` starts at
    int(round((H - n_px) / 2.0)), int(round((W - n_px) / 2.0)) (transforms/functional.py `center_crop`); `ToTensor`
    divides the bytes by 255; `Normalize` is (x - mean) / std in fp32.  torchvision is absent from this image, so these
    four size / rounding rules are restated from its published source.
  * Pillow (`Image.resize(..., BICUBIC)` -> libImaging/Resample.c `ImagingResample`): a separable two-pass convolution,
    horizontal then vertical, on 8-bit channels with 22-bit fixed-point coefficients and an 8-bit clipped intermediate.
    Pillow 12.2.0 IS installed here: tools/make_golden_preproc.py runs it to pin this restatement
    (tests/golden/preproc_*.npz), and tests/test_oracle.py re-checks the restatement against Pillow itself when it is
    importable.

Everything is integer arithmetic after the coefficients are fixed, so the parity bar is bit-exact.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients are scaled by 1 << 22
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    """Resample.c `bicubic_filter`, a = -0.5, support 2."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Resample.c `precompute_coeffs` + `normalize_coeffs_8bpc` for the box (0, in_size).
    Returns bounds int32 [out, 2] = (first source index, tap count) and integer coefficients int32 [out, ksize]."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (double)(in1 - in0) / outSize, box is float
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)             # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """`Image.resize((out_w, out_h), BICUBIC)` of a uint8 [H, W, C] array (ImagingResample, 8 bits per channel)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w, _ = img.shape
    cur = img
    bv, kv = precompute_coeffs(h, out_h)
    if out_w != w:                                    # horizontal pass, only on the rows the vertical pass reads
        bh, kh = precompute_coeffs(w, out_w)
        first, last = int(bv[0, 0]), int(bv[-1, 0] + bv[-1, 1])
        rows = cur[first:last].astype(np.int64)
        tmp = np.empty((last - first, out_w, cur.shape[2]), np.uint8)
        for xx in range(out_w):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = (rows[:, x0:x0 + n, :] * kh[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
        bv = bv.copy()
        bv[:, 0] -= first
    if out_h != h:                                    # vertical pass
        src = cur.astype(np.int64)
        out = np.empty((out_h, cur.shape[1], cur.shape[2]), np.uint8)
        for yy in range(out_h):
            y0, n = int(bv[yy, 0]), int(bv[yy, 1])
            acc = (src[y0:y0 + n] * kv[yy, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[yy] = _clip8(acc)
        cur = out
    return cur if cur is not img else img.copy()


def resized_size(w: int, h: int, n_px: int) -> Tuple[int, int]:
    """torchvision `Resize(int)`: (new_w, new_h); short side -> n_px, long side -> int(n_px * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == n_px:
        return w, h
    new_short, new_long = n_px, int(n_px * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def crop_origin(w: int, h: int, n_px: int) -> Tuple[int, int]:
    """torchvision `center_crop`: (left, top) with Python's round-half-even."""
    return int(round((w - n_px) / 2.0)), int(round((h - n_px) / 2.0))


def transform_u8(img: np.ndarray, n_px: int) -> np.ndarray:
    """Resize(n_px, BICUBIC) + CenterCrop(n_px) on a uint8 RGB [H, W, 3] array -> uint8 [n_px, n_px, 3]."""
    h, w, _ = img.shape
    nw, nh = resized_size(w, h, n_px)
    r = resize_bicubic(img, nw, nh) if (nw, nh) != (w, h) else img
    left, top = crop_origin(nw, nh, n_px)
    return np.ascontiguousarray(r[top:top + n_px, left:left + n_px])


def normalize(u8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize: uint8 [R, R, 3] -> fp32 [3, R, R], every step in fp32 like the reference's tensors."""
    t = np.transpose(u8, (2, 0, 1)).astype(np.float32) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)
    return ((t - mean) / std).astype(np.float32)


def transform(img: np.ndarray, n_px: int) -> np.ndarray:
    return normalize(transform_u8(img, n_px))
