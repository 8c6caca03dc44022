"""Device-side image transform: Resize(n_px, BICUBIC) + CenterCrop(n_px) + ToTensor + Normalize, bit-exact with the
reference's torchvision / Pillow pipeline (clip/clip.py:71-78 = dataset/imagenet_group.py:27-34).

The host decodes (PIL) and hands over raw RGB bytes of whatever size the files have; everything after the decode runs in
`hgr_preprocess_bicubic` on the GPU.  What stays on the host is the part of Pillow's `precompute_coeffs` that is
double-precision: the bicubic taps of every output row / column.  They depend only on (input size, output size), are
computed here with the same operations in the same order as libImaging/Resample.c (vectorised over the output index)
and cached per image size - ImageNet has a few dozen distinct sizes per thousand files.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

PRECISION_BITS = 22
MAX_TAPS = 96          # taps per output pixel the kernel tables are allowed to carry (downscale factor <= ~23)


def resized_size(w: int, h: int, n_px: int) -> Tuple[int, int]:
    """torchvision `Resize(int)` on a PIL image: the short side becomes n_px, the long side int(n_px * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    if short == n_px:
        return w, h
    new_short, new_long = n_px, int(n_px * long / short)
    return (new_short, new_long) if w <= h else (new_long, new_short)


def crop_origin(w: int, h: int, n_px: int) -> Tuple[int, int]:
    """torchvision `center_crop`: (left, top), Python round (half to even)."""
    return int(round((w - n_px) / 2.0)), int(round((h - n_px) / 2.0))


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@lru_cache(maxsize=4096)
def resize_taps(in_size: int, out_size: int, start: int, count: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (bicubic, box = the whole axis) for output indices
    [start, start + count): bounds int32 [count, 2] = (first source index, taps), taps int32 [count, ksize]."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(start, start + count, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    t = np.arange(ksize, dtype=np.int64)[None, :]
    live = t < xmax[:, None]
    w = _bicubic(((t + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where(live, w, 0.0)
    ww = np.zeros(count, np.float64)
    for i in range(ksize):                      # sequential sum, tap order: the same rounding as the C loop
        ww = ww + w[:, i]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    fixed = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS)))
    taps = np.where(live, fixed, 0.0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    taps.setflags(write=False); bounds.setflags(write=False)
    return bounds, taps


@lru_cache(maxsize=4096)
def image_tables(w: int, h: int, n_px: int):
    """(xb, xk, yb, yk) of one image size for the cropped n_px x n_px output."""
    nw, nh = resized_size(w, h, n_px)
    left, top = crop_origin(nw, nh, n_px)
    xb, xk = resize_taps(w, nw, left, n_px)
    yb, yk = resize_taps(h, nh, top, n_px)
    return xb, xk, yb, yk


class BatchPreprocessor:
    """Packs decoded images, uploads them with their tap tables and runs the transform kernel.

    >>> pre = BatchPreprocessor(224, "cuda:0")
    >>> x = pre(list_of_uint8_hwc_arrays)                  # fp32 [B, 3, 224, 224], what the reference's loader yields
    >>> x = pre(list_of_uint8_hwc_arrays, output="u8")     # uint8 [B, 224, 224, 3] for CLIP.encode_image's fused path
    """

    def __init__(self, n_px: int, device, mean: Sequence[float] = ops.CLIP_MEAN, std: Sequence[float] = ops.CLIP_STD):
        self.n_px, self.device = int(n_px), torch.device(device)
        self.mean, self.std = tuple(mean), tuple(std)
        self._pin: Optional[torch.Tensor] = None
        self._threads = None

    def _pool(self):
        if self._threads is None:
            from concurrent.futures import ThreadPoolExecutor
            self._threads = ThreadPoolExecutor(8)
        return self._threads

    def _pinned(self, nbytes: int) -> torch.Tensor:
        if self._pin is None or self._pin.numel() < nbytes:
            self._pin = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        return self._pin

    def __call__(self, images: List[np.ndarray], output: str = "f32") -> torch.Tensor:
        assert output in ("f32", "u8") and len(images) > 0
        r, b = self.n_px, len(images)
        # one set of tap tables per distinct image size (a batch of ImageNet files has a handful)
        sizes_hw, tab = {}, np.empty(b, np.int32)
        for i, im in enumerate(images):
            if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                raise ValueError(f"expected uint8 [H, W, 3] RGB arrays, got {im.dtype} {im.shape}")
            tab[i] = sizes_hw.setdefault((im.shape[0], im.shape[1]), len(sizes_hw))
        tabs = [image_tables(w, h, r) for (h, w) in sizes_hw]
        t = len(tabs)
        kx, ky = max(x[1].shape[1] for x in tabs), max(x[3].shape[1] for x in tabs)
        kx = (kx + 3) // 4 * 4                                   # the kernel consumes 4 horizontal taps per step
        if max(kx, ky) > MAX_TAPS:
            raise ValueError(f"an image is more than {MAX_TAPS // 4 - 1}x larger than the crop: reduce it on the host first")
        nbytes_img = np.array([im.shape[0] * im.shape[1] * 3 for im in images], np.int64)
        off = np.zeros(b, np.int64)
        off[1:] = np.cumsum((nbytes_img[:-1] + 15) // 16 * 16)                 # 16-byte aligned starts
        total = int(off[-1] + nbytes_img[-1])
        # one pinned staging buffer: [image bytes | off | hw | tab | xb | yb | (pad) xk | yk], one H2D copy
        n_i32 = b * 2 + b + 2 * t * r * 2
        n_i32 = (n_i32 + 3) // 4 * 4                             # xk starts 16-byte aligned
        q_xk = n_i32
        n_i32 += t * r * kx + t * r * ky
        base_meta = (total + 15) // 16 * 16
        p_i32 = base_meta + (b * 8 + 15) // 16 * 16
        nbytes = p_i32 + n_i32 * 4 + 32                           # + slack: a row's bytes are fetched 12 at a time
        pin = self._pinned(nbytes)
        host = pin.numpy()

        def put(k):                                              # numpy copies release the GIL: pack with a few threads
            host[off[k]:off[k] + nbytes_img[k]] = np.ascontiguousarray(images[k]).reshape(-1)
        if total > (8 << 20):
            list(self._pool().map(put, range(b)))
        else:
            for k in range(b):
                put(k)
        host[base_meta:base_meta + b * 8].view(np.int64)[:] = off
        i32 = host[p_i32:p_i32 + n_i32 * 4].view(np.int32)
        q = 0
        i32[q:q + 2 * b].reshape(b, 2)[:] = [im.shape[:2] for im in images]; q_hw = q; q += 2 * b
        i32[q:q + b] = tab; q_tab = q; q += b
        xb = i32[q:q + t * r * 2].reshape(t, r, 2); q_xb = q; q += t * r * 2
        yb = i32[q:q + t * r * 2].reshape(t, r, 2); q_yb = q; q += t * r * 2
        xk = i32[q_xk:q_xk + t * r * kx].reshape(t, r, kx)
        q_yk = q_xk + t * r * kx
        yk = i32[q_yk:q_yk + t * r * ky].reshape(t, r, ky)
        xk[:] = 0; yk[:] = 0
        for i, x in enumerate(tabs):
            xb[i] = x[0]; xk[i, :, :x[1].shape[1]] = x[1]
            yb[i] = x[2]; yk[i, :, :x[3].shape[1]] = x[3]
        dev = pin[:nbytes].to(self.device, non_blocking=True)
        d_i32 = dev[p_i32:p_i32 + n_i32 * 4].view(torch.int32)
        out_u8 = torch.empty((b, r, r, 3), dtype=torch.uint8, device=self.device) if output == "u8" else None
        out_f32 = torch.empty((b, 3, r, r), dtype=torch.float32, device=self.device) if output == "f32" else None
        ops.preprocess_bicubic(dev, dev[base_meta:base_meta + b * 8].view(torch.int64), d_i32[q_hw:q_hw + 2 * b], d_i32[q_tab:q_tab + b],
                               d_i32[q_xb:q_xb + t * r * 2], d_i32[q_xk:q_xk + t * r * kx], kx,
                               d_i32[q_yb:q_yb + t * r * 2], d_i32[q_yk:q_yk + t * r * ky], ky,
                               out_u8, out_f32, self.mean, self.std, b, r)
        # the staging buffer is reused by the next call: the copy must have been consumed before we return to the host
        torch.cuda.current_stream(self.device).synchronize()
        return out_f32 if output == "f32" else out_u8
