"""Deterministic synthetic inputs for the HGR-Net zero-shot path.

Nothing the reference needs at run time exists offline (no pretrained CLIP weights, no
ImageNet-21K hierarchy file, no WordNet, no images - SURVEY.md F9), so every benchmark and
parity test runs on data produced here.  All values come from a counter-based integer hash
(splitmix64 over element index, keyed by ``(seed, tensor name)``) turned into float64 and then
cast, so the same call yields the same bits in the build container and on the GPU box without
any fixture travelling.

Shapes and key names follow the reference's CLIP ``state_dict`` schema
(reference ``clip/model.py:239-368`` for the module tree, ``:395-432`` for how the architecture is
inferred back from the shapes).  Initial scales follow ``clip/model.py:295-322`` except that
``bn3.weight`` is NOT zeroed (a zero ``bn3`` would make every ResNet block an identity and the
parity tests would exercise nothing) and LayerNorm/BatchNorm affine terms are perturbed away from
(1, 0) so that they matter.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

_U64 = np.uint64
_MASK = (1 << 64) - 1


def fnv1a64(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _U64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        return z ^ (z >> _U64(31))


def _key(seed: int, name: str, stream: int = 0) -> np.uint64:
    k = (fnv1a64(name) ^ ((seed * 0xD1342543DE82EF95) & _MASK) ^ ((stream * 0xA0761D6478BD642F) & _MASK)) & _MASK
    return _U64(k)


def uniform(seed: int, name: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in [0, 1)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix64(_splitmix64(idx ^ _key(seed, name, stream)) + idx)
    return (h >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, name: str, n: int) -> np.ndarray:
    """n float64 standard-normal values (Box-Muller on two hash streams)."""
    u1 = uniform(seed, name, n, 1)
    u2 = uniform(seed, name, n, 2)
    return np.sqrt(-2.0 * np.log1p(-u1)) * np.cos(2.0 * math.pi * u2)


def randint(seed: int, name: str, n: int, lo: int, hi: int) -> np.ndarray:
    """n int64 values in [lo, hi)."""
    return (lo + np.floor(uniform(seed, name, n) * (hi - lo))).astype(np.int64)


def _t(a: np.ndarray, shape: Sequence[int]) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32).reshape(tuple(shape))))


# --------------------------------------------------------------------------------------------
# CLIP configurations (dims probed from the reference's CLIP class)
# --------------------------------------------------------------------------------------------
CLIP_CONFIGS: Dict[str, dict] = {
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=32, context_length=77, vocab_size=49408,
                     transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=16, context_length=77, vocab_size=49408,
                     transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024,
                     vision_patch_size=14, context_length=77, vocab_size=49408,
                     transformer_width=768, transformer_heads=12, transformer_layers=12),
    "RN50": dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3), vision_width=64,
                 vision_patch_size=None, context_length=77, vocab_size=49408,
                 transformer_width=512, transformer_heads=8, transformer_layers=12),
    "RN101": dict(embed_dim=512, image_resolution=224, vision_layers=(3, 4, 23, 3), vision_width=64,
                  vision_patch_size=None, context_length=77, vocab_size=49408,
                  transformer_width=512, transformer_heads=8, transformer_layers=12),
    "RN50x4": dict(embed_dim=640, image_resolution=288, vision_layers=(4, 6, 10, 6), vision_width=80,
                   vision_patch_size=None, context_length=77, vocab_size=49408,
                   transformer_width=640, transformer_heads=10, transformer_layers=12),
    "RN50x16": dict(embed_dim=768, image_resolution=384, vision_layers=(6, 8, 18, 8), vision_width=96,
                    vision_patch_size=None, context_length=77, vocab_size=49408,
                    transformer_width=768, transformer_heads=12, transformer_layers=12),
    # small configs for fast parity tests (heads must keep d_head = 64, clip/model.py:259,268,417)
    "tiny-vit": dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128,
                     vision_patch_size=32, context_length=77, vocab_size=512,
                     transformer_width=64, transformer_heads=1, transformer_layers=2),
    "small-vit": dict(embed_dim=128, image_resolution=96, vision_layers=3, vision_width=256,
                      vision_patch_size=32, context_length=77, vocab_size=1024,
                      transformer_width=128, transformer_heads=2, transformer_layers=3),
    # width 64 = RN50 / RN101's; the training tower (training_rn.py) needs a power-of-two width >= 64
    "tiny-rn": dict(embed_dim=64, image_resolution=64, vision_layers=(1, 1, 1, 1), vision_width=64,
                    vision_patch_size=None, context_length=77, vocab_size=512,
                    transformer_width=64, transformer_heads=1, transformer_layers=2),
    # non-power-of-two width (RN50x4 / RN50x16 style): planes 48 / 96 are stored padded to 64 / 128
    "small-rnx": dict(embed_dim=128, image_resolution=96, vision_layers=(1, 2, 1, 1), vision_width=48,
                      vision_patch_size=None, context_length=77, vocab_size=512,
                      transformer_width=64, transformer_heads=1, transformer_layers=2),
    "small-rn": dict(embed_dim=128, image_resolution=96, vision_layers=(2, 1, 2, 1), vision_width=64,
                     vision_patch_size=None, context_length=77, vocab_size=512,
                     transformer_width=64, transformer_heads=1, transformer_layers=2),
}


def _ln(sd, seed, prefix, width):
    sd[prefix + ".weight"] = _t(1.0 + 0.1 * normal(seed, prefix + ".weight", width), [width])
    sd[prefix + ".bias"] = _t(0.05 * normal(seed, prefix + ".bias", width), [width])


def _bn(sd, seed, prefix, ch, gain=1.0):
    sd[prefix + ".weight"] = _t(gain * (1.0 + 0.1 * normal(seed, prefix + ".weight", ch)), [ch])
    sd[prefix + ".bias"] = _t(0.05 * normal(seed, prefix + ".bias", ch), [ch])
    sd[prefix + ".running_mean"] = _t(0.1 * normal(seed, prefix + ".running_mean", ch), [ch])
    sd[prefix + ".running_var"] = _t(0.5 + uniform(seed, prefix + ".running_var", ch), [ch])
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def _resblocks(sd, seed, prefix, width, layers):
    attn_std = width ** -0.5
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        sd[p + ".attn.in_proj_weight"] = _t(attn_std * normal(seed, p + ".attn.in_proj_weight", 3 * width * width), [3 * width, width])
        sd[p + ".attn.in_proj_bias"] = _t(0.02 * normal(seed, p + ".attn.in_proj_bias", 3 * width), [3 * width])
        sd[p + ".attn.out_proj.weight"] = _t(proj_std * normal(seed, p + ".attn.out_proj.weight", width * width), [width, width])
        sd[p + ".attn.out_proj.bias"] = _t(0.02 * normal(seed, p + ".attn.out_proj.bias", width), [width])
        _ln(sd, seed, p + ".ln_1", width)
        sd[p + ".mlp.c_fc.weight"] = _t(fc_std * normal(seed, p + ".mlp.c_fc.weight", 4 * width * width), [4 * width, width])
        sd[p + ".mlp.c_fc.bias"] = _t(0.02 * normal(seed, p + ".mlp.c_fc.bias", 4 * width), [4 * width])
        sd[p + ".mlp.c_proj.weight"] = _t(proj_std * normal(seed, p + ".mlp.c_proj.weight", 4 * width * width), [width, 4 * width])
        sd[p + ".mlp.c_proj.bias"] = _t(0.02 * normal(seed, p + ".mlp.c_proj.bias", width), [width])
        _ln(sd, seed, p + ".ln_2", width)


def _conv(sd, seed, name, cout, cin, k):
    std = math.sqrt(2.0 / (cin * k * k))
    sd[name] = _t(std * normal(seed, name, cout * cin * k * k), [cout, cin, k, k])


def clip_state_dict(config: str | dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """fp32 CLIP state_dict with the reference's key schema (clip/model.py:239-368)."""
    cfg = CLIP_CONFIGS[config] if isinstance(config, str) else config
    sd: Dict[str, torch.Tensor] = {}
    wt, d = cfg["transformer_width"], cfg["embed_dim"]
    ctx, vocab = cfg["context_length"], cfg["vocab_size"]
    sd["positional_embedding"] = _t(0.01 * normal(seed, "positional_embedding", ctx * wt), [ctx, wt])
    sd["text_projection"] = _t(wt ** -0.5 * normal(seed, "text_projection", wt * d), [wt, d])
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07), dtype=torch.float32)
    sd["token_embedding.weight"] = _t(0.02 * normal(seed, "token_embedding.weight", vocab * wt), [vocab, wt])
    _ln(sd, seed, "ln_final", wt)
    _resblocks(sd, seed, "transformer", wt, cfg["transformer_layers"])

    vl, vw, res = cfg["vision_layers"], cfg["vision_width"], cfg["image_resolution"]
    if isinstance(vl, (tuple, list)):
        _conv(sd, seed, "visual.conv1.weight", vw // 2, 3, 3)
        _bn(sd, seed, "visual.bn1", vw // 2)
        _conv(sd, seed, "visual.conv2.weight", vw // 2, vw // 2, 3)
        _bn(sd, seed, "visual.bn2", vw // 2)
        _conv(sd, seed, "visual.conv3.weight", vw, vw // 2, 3)
        _bn(sd, seed, "visual.bn3", vw)
        inplanes = vw
        for li, nblocks in enumerate(vl):
            planes = vw * (2 ** li)
            stride = 1 if li == 0 else 2
            for j in range(nblocks):
                p = f"visual.layer{li + 1}.{j}"
                _conv(sd, seed, p + ".conv1.weight", planes, inplanes, 1)
                _bn(sd, seed, p + ".bn1", planes)
                _conv(sd, seed, p + ".conv2.weight", planes, planes, 3)
                _bn(sd, seed, p + ".bn2", planes)
                _conv(sd, seed, p + ".conv3.weight", planes * 4, planes, 1)
                _bn(sd, seed, p + ".bn3", planes * 4, gain=0.25)   # keeps 16 residual adds O(1)
                if j == 0 and (stride > 1 or inplanes != planes * 4):
                    _conv(sd, seed, p + ".downsample.0.weight", planes * 4, inplanes, 1)
                    _bn(sd, seed, p + ".downsample.1", planes * 4)
                inplanes = planes * 4
        e = vw * 32
        sp = res // 32
        std = e ** -0.5
        sd["visual.attnpool.positional_embedding"] = _t(std * normal(seed, "visual.attnpool.positional_embedding", (sp * sp + 1) * e), [sp * sp + 1, e])
        for nm, out in (("k_proj", e), ("q_proj", e), ("v_proj", e), ("c_proj", d)):
            p = f"visual.attnpool.{nm}"
            sd[p + ".weight"] = _t(std * normal(seed, p + ".weight", out * e), [out, e])
            sd[p + ".bias"] = _t(0.02 * normal(seed, p + ".bias", out), [out])
    else:
        ps = cfg["vision_patch_size"]
        grid = res // ps
        scale = vw ** -0.5
        sd["visual.class_embedding"] = _t(scale * normal(seed, "visual.class_embedding", vw), [vw])
        sd["visual.positional_embedding"] = _t(scale * normal(seed, "visual.positional_embedding", (grid * grid + 1) * vw), [grid * grid + 1, vw])
        sd["visual.proj"] = _t(scale * normal(seed, "visual.proj", vw * d), [vw, d])
        sd["visual.conv1.weight"] = _t((3 * ps * ps) ** -0.5 * normal(seed, "visual.conv1.weight", vw * 3 * ps * ps), [vw, 3, ps, ps])
        _ln(sd, seed, "visual.ln_pre", vw)
        _ln(sd, seed, "visual.ln_post", vw)
        _resblocks(sd, seed, "visual.transformer", vw, vl)
    # Published CLIP checkpoints hold fp16 values, and the reference's loader rounds every
    # Linear/Conv/attention weight to fp16 anyway (build_model: convert_weights then load_state_dict,
    # clip/model.py:430-431).  Making all synthetic values fp16-representable keeps that step a no-op,
    # so "same weights" means the same bits on both sides.
    for k, v in sd.items():
        if v.dtype == torch.float32 and k != "logit_scale":
            sd[k] = v.to(torch.float16).to(torch.float32)
    return sd


def images(batch: int, resolution: int = 224, seed: int = 1234) -> torch.Tensor:
    """fp32 [batch, 3, R, R] ~ N(0, 1): stands in for normalised 224x224 crops (main.py:133)."""
    n = batch * 3 * resolution * resolution
    return _t(normal(seed, "images", n), [batch, 3, resolution, resolution])


# --------------------------------------------------------------------------------------------
# Synthetic WordNet-like DAG, prompts and splits
# --------------------------------------------------------------------------------------------
ROOT = "fall11"  # the reference's root wnid (utils.py:45-46)

# relative node mass per depth 0..11: peaked at layers 2-6 (supplementary PDF p.3 Fig.1)
_LAYER_MASS = [0.015, 0.06, 0.13, 0.19, 0.20, 0.16, 0.11, 0.07, 0.035, 0.017, 0.008, 0.005]


def make_dag(n_nodes: int, depth: int = 12, seed: int = 7, multi_parent: float = 0.03) -> List[List[str]]:
    """Edge list ``[[parent_wnid, child_wnid], ...]`` in the format ``gen_tree`` reads (utils.py:40-43).

    Nodes get wnids ``n%08d``; depth-0 nodes hang off ``fall11``.  A fraction ``multi_parent`` of
    the nodes gets a second parent one or two layers up, so shortest-path depth and BFS tie-breaking
    are exercised.  Edges are emitted in a seeded shuffled order, so node ids (first appearance in
    the edge list, utils.py:44) are not sorted by depth.
    """
    depth = max(1, min(depth, len(_LAYER_MASS), n_nodes))
    mass = np.array(_LAYER_MASS[:depth], dtype=np.float64)
    counts = np.maximum(1, np.floor(mass / mass.sum() * n_nodes)).astype(np.int64)
    while counts.sum() > n_nodes:
        counts[int(np.argmax(counts))] -= 1
    counts[int(np.argmax(mass))] += n_nodes - counts.sum()
    layers: List[List[int]] = []
    nid = 0
    for c in counts:
        layers.append(list(range(nid, nid + int(c))))
        nid += int(c)
    wn = lambda i: "n%08d" % (i + 1)
    edges: List[Tuple[str, str]] = []
    for lvl, nodes in enumerate(layers):
        if lvl == 0:
            edges += [(ROOT, wn(i)) for i in nodes]
            continue
        up = layers[lvl - 1]
        pick = randint(seed, f"dag.parent.{lvl}", len(nodes), 0, len(up))
        extra = uniform(seed, f"dag.extra.{lvl}", len(nodes))
        hop = randint(seed, f"dag.hop.{lvl}", len(nodes), 1, 3)
        pick2 = uniform(seed, f"dag.parent2.{lvl}", len(nodes))
        for j, i in enumerate(nodes):
            edges.append((wn(up[int(pick[j])]), wn(i)))
            if extra[j] < multi_parent:
                l2 = max(0, lvl - int(hop[j]))
                cand = layers[l2]
                p2 = cand[int(pick2[j] * len(cand))]
                if wn(p2) != edges[-1][0]:
                    edges.append((wn(p2), wn(i)))
    # root edges first keeps ``fall11`` as the first node networkx sees, like a real dump; the rest
    # is shuffled so that ids do not follow depth order
    root_e = [e for e in edges if e[0] == ROOT]
    rest = [e for e in edges if e[0] != ROOT]
    order = np.argsort(uniform(seed, "dag.shuffle", len(rest)), kind="stable")
    return [list(e) for e in root_e] + [list(rest[int(k)]) for k in order]


SOT, EOT = 49406, 49407  # clip/simple_tokenizer.py special ids (used when vocab is the real 49408)


def make_tokens(n: int, seed: int = 11, vocab_size: int = 49408, context_length: int = 77, n_ctx: int = 0) -> torch.Tensor:
    """int64 [n, 77] token ids shaped like ``clip.tokenize('a photo of a {name}.')`` (clip.py:188-224).

    SOT, 4 template ids, 1-4 name ids, '.', EOT, zero padding.  EOT is the largest id of every row
    because ``encode_text`` locates it with ``argmax`` (clip/model.py:350).
    """
    sot, eot = (SOT, EOT) if vocab_size >= 49408 else (vocab_size - 2, vocab_size - 1)
    hi = sot  # ordinary ids are < SOT
    tmpl = [min(320, hi - 1), min(1125, hi - 2), min(539, hi - 3), min(320, hi - 1)]  # 'a photo of a'
    if n_ctx:
        tmpl = [min(343, hi - 5)] * n_ctx                       # CoOp: n_ctx 'X' placeholders (model/CoOp.py:70)
    dot = min(269, hi - 4)
    name_len = randint(seed, "tok.len", n, 1, 5)
    name_ids = randint(seed, "tok.ids", n * 4, 1, hi).reshape(n, 4)
    out = np.zeros((n, context_length), dtype=np.int64)
    for i in range(n):
        row = [sot] + tmpl + [int(t) for t in name_ids[i, : int(name_len[i])]] + [dot, eot]
        out[i, : len(row)] = row
    return torch.from_numpy(out)


def make_splits(nodes: Sequence[str], leaf_mask: Sequence[bool], n_train: int, n_test: int, seed: int = 13) -> Dict[str, List[str]]:
    """Seeded seen/unseen split in the ``splits_for_tree.json`` format main.py:227-229 reads.

    ``all`` = every node (``--model_train all``), ``train`` = seen classes, ``rest`` = unseen classes
    (``--model_test rest``); unseen classes are drawn from leaves first, like ImageNet-21K's.
    """
    n = len(nodes)
    order = np.argsort(uniform(seed, "split.perm", n), kind="stable")
    leaves = [int(i) for i in order if leaf_mask[int(i)]]
    inner = [int(i) for i in order if not leaf_mask[int(i)]]
    pool = leaves + inner
    test = pool[:n_test]
    train = pool[n_test:n_test + n_train]
    return {"all": list(nodes), "train": [nodes[i] for i in train], "rest": [nodes[i] for i in test]}
