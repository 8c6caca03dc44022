"""``clip.load`` / ``clip.tokenize`` with the reference's call surface (clip/clip.py:86-224).

Host-side only.  There is no network here, so a model *name* resolves to ``{download_root}/{file}``
if that file already exists and otherwise raises; a path to a checkpoint (plain ``state_dict`` or a
TorchScript archive, clip/clip.py:118-128) always works.
"""
from __future__ import annotations

import os
from typing import List, Union

import torch

from .model import build_model

# file names the reference downloads (clip/clip.py:25-32); ViT-L/14 has no entry there either
_MODELS = {"RN50": "RN50.pt", "RN101": "RN101.pt", "RN50x4": "RN50x4.pt", "RN50x16": "RN50x16.pt",
           "ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt"}

__all__ = ["available_models", "load", "tokenize"]


def available_models() -> List[str]:
    return list(_MODELS.keys())


def _transform(n_px: int):
    """Bicubic resize -> center crop -> RGB -> [0,1] tensor -> CLIP mean/std (clip/clip.py:71-78): the per-image host
    callable `clip.load` returns, like the reference's.  Batches go through hgr_net_amd.preprocess.BatchPreprocessor
    (the same arithmetic on the GPU, bit-exact)."""
    mean = torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)
    std = torch.tensor((0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)

    def apply(image):
        import numpy as np
        from PIL import Image
        from ..preprocess import crop_origin, resized_size         # torchvision's Resize(int) / CenterCrop size rules
        w, h = image.size
        if (w, h) != resized_size(w, h, n_px):
            image = image.resize(resized_size(w, h, n_px), Image.BICUBIC)
        w, h = image.size
        left, top = crop_origin(w, h, n_px)
        image = image.crop((left, top, left + n_px, top + n_px)).convert("RGB")
        t = torch.from_numpy(np.asarray(image, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (t - mean) / std

    return apply


def load(name: str, device: Union[str, torch.device, int] = "cuda", jit: bool = False, download_root: str = None,
         image_dtype: str = "f16", text_dtype: str = "f16"):
    """Returns ``(model, preprocess)`` like the reference.  ``device`` may be an int CUDA ordinal
    (the reference passes ``opts.device`` raw, model/clip_tree.py:23,34)."""
    if os.path.isfile(name):
        path = name
    elif name in _MODELS:
        path = os.path.join(download_root or os.path.expanduser("~/.cache/clip"), _MODELS[name])
        if not os.path.isfile(path):
            raise RuntimeError(f"Model {name}: {path} not present and downloading is not possible here; "
                               f"pass the path of a checkpoint instead")
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    if jit:
        raise RuntimeError("jit=True is not supported: the forward path is libhgr.so, not TorchScript")
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
    if isinstance(device, int):
        device = f"cuda:{device}"
    model = build_model(sd, image_dtype=image_dtype, text_dtype=text_dtype).to(device)
    return model, _transform(model.visual.input_resolution)


_tokenizer = None


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.LongTensor:
    """int64 [n, context_length]: SOT + BPE ids + EOT, zero padded; RuntimeError when too long
    (clip/clip.py:188-224).  Needs the BPE merges file (``HGR_BPE_VOCAB`` or next to a reference checkout)."""
    global _tokenizer
    if _tokenizer is None:
        from .simple_tokenizer import SimpleTokenizer
        _tokenizer = SimpleTokenizer()
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = _tokenizer.encoder["<|startoftext|>"], _tokenizer.encoder["<|endoftext|>"]
    result = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, text in enumerate(texts):
        toks = [sot] + _tokenizer.encode(text) + [eot]
        if len(toks) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {text} is too long for context length {context_length}")
            toks = toks[:context_length]
            toks[-1] = eot
        result[i, : len(toks)] = torch.tensor(toks)
    return result
