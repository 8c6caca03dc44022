"""ORACLE (test infrastructure, never a product path): CPU fp32 restatement of the CLIP towers.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  It restates, as plain functions over a ``state_dict``, the arithmetic of the reference's
``clip/model.py``; every function cites the lines it follows.  It is checked against an import of the
real reference by ``tools/make_golden.py`` (run in the build container, where ``/root/reference``
exists) and against the committed fixtures in ``tests/golden/`` by ``tests/test_oracle.py``.

Parity status: the reference ships no tests or golden vectors of its own, and the
attention / conv / layer-norm arithmetic lives in PyTorch, not in the reference.  The
oracle is therefore pinned by fixtures generated from the reference's own code run here, with the
generating script committed (tools/make_golden.py).

``rd`` hooks: every matmul-shaped product takes its two operands through ``rd`` first.  With the
default identity this is the fp32 reference arithmetic; with ``round_bf16`` it emulates the rounding
points of the HIP path (bf16 MFMA inputs, fp32 accumulate, fp32 residual stream / LayerNorm / softmax)
and is used to budget the 1e-3 logit tolerance before it is measured on hardware.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
Rd = Callable[[torch.Tensor], torch.Tensor]


def identity(x: torch.Tensor) -> torch.Tensor:
    return x


def round_bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def round_f16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float16).to(torch.float32)


def infer_config(sd: SD) -> dict:
    """Architecture from tensor shapes, as ``build_model`` does (clip/model.py:395-418)."""
    vit = "visual.proj" in sd
    if vit:
        vw = sd["visual.conv1.weight"].shape[0]
        vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        ps = sd["visual.conv1.weight"].shape[-1]
        grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        res = ps * grid
    else:
        vl = tuple(len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}")}) for b in (1, 2, 3, 4))
        vw = sd["visual.layer1.0.conv1.weight"].shape[0]
        ow = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        ps = None
        res = ow * 32
    wt = sd["ln_final.weight"].shape[0]
    return dict(embed_dim=sd["text_projection"].shape[1], image_resolution=res, vision_layers=vl,
                vision_width=vw, vision_patch_size=ps, context_length=sd["positional_embedding"].shape[0],
                vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=wt,
                transformer_heads=wt // 64,
                transformer_layers=len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")}))


def layer_norm(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    """fp32 LayerNorm, eps 1e-5, affine (clip/model.py:153-159)."""
    return F.layer_norm(x.float(), (x.shape[-1],), sd[prefix + ".weight"].float(), sd[prefix + ".bias"].float(), 1e-5)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x) (clip/model.py:162-164)."""
    return x * torch.sigmoid(1.702 * x)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], rd: Rd) -> torch.Tensor:
    y = rd(x) @ rd(w.float()).t()
    return y if b is None else y + b.float()


def mha(x: torch.Tensor, sd: SD, prefix: str, heads: int, causal: bool, rd: Rd) -> torch.Tensor:
    """Self-attention of ``nn.MultiheadAttention`` as the reference calls it (clip/model.py:171,183):
    packed in-projection, per-head softmax(q k^T / sqrt(d) + mask) v, out-projection.
    x: [B, L, W] (batch-major here; the reference's [L, B, W] is the same arithmetic)."""
    b, l, w = x.shape
    d = w // heads
    qkv = linear(x, sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"], rd)
    qkv = rd(qkv)                                       # HIP path stores packed QKV in 16 bit
    q, k, v = qkv.split(w, dim=-1)
    q = q.reshape(b, l, heads, d).transpose(1, 2)
    k = k.reshape(b, l, heads, d).transpose(1, 2)
    v = v.reshape(b, l, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (d ** -0.5)
    if causal:                                           # build_attention_mask, clip/model.py:324-330
        s = s + torch.full((l, l), float("-inf")).triu_(1)
    p = torch.softmax(s, dim=-1)
    o = (rd(p) @ v).transpose(1, 2).reshape(b, l, w)
    return linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"], rd)


def resblock(x: torch.Tensor, sd: SD, prefix: str, heads: int, causal: bool, rd: Rd) -> torch.Tensor:
    """x + attn(ln_1 x); x + c_proj(QuickGELU(c_fc(ln_2 x)))  (clip/model.py:185-188)."""
    x = x + mha(layer_norm(x, sd, prefix + ".ln_1"), sd, prefix + ".attn", heads, causal, rd)
    h = linear(layer_norm(x, sd, prefix + ".ln_2"), sd[prefix + ".mlp.c_fc.weight"], sd[prefix + ".mlp.c_fc.bias"], rd)
    return x + linear(quick_gelu(h), sd[prefix + ".mlp.c_proj.weight"], sd[prefix + ".mlp.c_proj.bias"], rd)


def transformer(x, sd, prefix, layers, heads, causal, rd, taps=None):
    for i in range(layers):
        x = resblock(x, sd, f"{prefix}.resblocks.{i}", heads, causal, rd)
        if taps is not None:
            taps[f"{prefix}.resblocks.{i}"] = x
    return x


def vit_forward(sd: SD, image: torch.Tensor, rd: Rd = identity, taps: Optional[dict] = None) -> torch.Tensor:
    """VisionTransformer.forward (clip/model.py:219-236): patch conv (k = s = patch, no bias) ->
    [B, grid^2, W]; prepend class embedding; + positional; ln_pre; blocks; ln_post(token 0); @ proj."""
    w = sd["visual.conv1.weight"].float()
    width, _, ps, _ = w.shape
    b, _, r, _ = image.shape
    g = r // ps
    # stride == kernel, so the conv is a GEMM over disjoint patches
    patches = image.float().reshape(b, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(b, g * g, 3 * ps * ps)
    x = rd(patches) @ rd(w.reshape(width, -1)).t()
    cls = sd["visual.class_embedding"].float().expand(b, 1, width)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"].float()
    x = layer_norm(x, sd, "visual.ln_pre")
    if taps is not None:
        taps["visual.ln_pre"] = x
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    x = transformer(x, sd, "visual.transformer", layers, width // 64, False, rd, taps)
    x = layer_norm(x[:, 0, :], sd, "visual.ln_post")
    return rd(x) @ rd(sd["visual.proj"].float())


def _bn(x, sd, prefix):
    """Inference BatchNorm with running stats (the model is always in eval(): clip_tree.py:46)."""
    return F.batch_norm(x, sd[prefix + ".running_mean"].float(), sd[prefix + ".running_var"].float(),
                        sd[prefix + ".weight"].float(), sd[prefix + ".bias"].float(), False, 0.0, 1e-5)


def _bottleneck(x, sd, p, stride):
    """Bottleneck.forward (clip/model.py:40-53)."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"].float()), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"].float(), padding=1), sd, p + ".bn2"))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"].float()), sd, p + ".bn3")
    idn = x
    if p + ".downsample.0.weight" in sd:
        idn = F.avg_pool2d(x, stride) if stride > 1 else x
        idn = _bn(F.conv2d(idn, sd[p + ".downsample.0.weight"].float()), sd, p + ".downsample.1")
    return F.relu(out + idn)


def attnpool(x, sd, heads):
    """AttentionPool2d.forward (clip/model.py:66-90): tokens = [mean, 7x7 cells] + pos; MHA with
    separate q/k/v projections; only token 0's output is returned, so only its query row matters."""
    b, c, h, w = x.shape
    t = x.reshape(b, c, h * w).permute(0, 2, 1)
    t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd["visual.attnpool.positional_embedding"].float()
    p = "visual.attnpool."
    q = F.linear(t[:, :1], sd[p + "q_proj.weight"].float(), sd[p + "q_proj.bias"].float())
    k = F.linear(t, sd[p + "k_proj.weight"].float(), sd[p + "k_proj.bias"].float())
    v = F.linear(t, sd[p + "v_proj.weight"].float(), sd[p + "v_proj.bias"].float())
    d = c // heads
    q = q.reshape(b, 1, heads, d).transpose(1, 2)
    k = k.reshape(b, -1, heads, d).transpose(1, 2)
    v = v.reshape(b, -1, heads, d).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1)
    o = (a @ v).transpose(1, 2).reshape(b, c)
    return F.linear(o, sd[p + "c_proj.weight"].float(), sd[p + "c_proj.bias"].float())


def rn_forward(sd: SD, image: torch.Tensor) -> torch.Tensor:
    """ModifiedResNet.forward (clip/model.py:135-150)."""
    x = image.float()
    x = F.relu(_bn(F.conv2d(x, sd["visual.conv1.weight"].float(), stride=2, padding=1), sd, "visual.bn1"))
    x = F.relu(_bn(F.conv2d(x, sd["visual.conv2.weight"].float(), padding=1), sd, "visual.bn2"))
    x = F.relu(_bn(F.conv2d(x, sd["visual.conv3.weight"].float(), padding=1), sd, "visual.bn3"))
    x = F.avg_pool2d(x, 2)
    for li in (1, 2, 3, 4):
        j = 0
        while f"visual.layer{li}.{j}.conv1.weight" in sd:
            x = _bottleneck(x, sd, f"visual.layer{li}.{j}", 2 if (li > 1 and j == 0) else 1)
            j += 1
    width = sd["visual.layer1.0.conv1.weight"].shape[0]
    return attnpool(x, sd, width * 32 // 64)


def encode_image(sd: SD, image: torch.Tensor, rd: Rd = identity, taps: Optional[dict] = None) -> torch.Tensor:
    """CLIP.encode_image (clip/model.py:336)."""
    if "visual.proj" in sd:
        return vit_forward(sd, image, rd, taps)
    return rn_forward(sd, image)


def encode_text(sd: SD, text: torch.Tensor, rd: Rd = identity, trim: bool = False, ctx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """CLIP.encode_text (clip/model.py:339-352): embedding gather + positional; causal blocks;
    ln_final; row at argmax(token id) (= EOT); @ text_projection.

    ``trim=True`` runs the blocks only on positions <= max(EOT): the mask is causal and only the EOT
    row is read, so later positions cannot influence the result."""
    eot = text.argmax(dim=-1)
    if trim:
        text = text[:, : int(eot.max()) + 1]
    l = text.shape[1]
    x = sd["token_embedding.weight"].float()[text]
    if ctx is not None:
        x = torch.cat([x[:, :1], ctx.float().unsqueeze(0).expand(x.shape[0], -1, -1), x[:, 1 + ctx.shape[0]:]], dim=1)
    x = x + sd["positional_embedding"].float()[:l]
    wt = x.shape[-1]
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    x = transformer(x, sd, "transformer", layers, wt // 64, True, rd)
    x = layer_norm(x, sd, "ln_final")
    x = x[torch.arange(x.shape[0]), eot]
    return rd(x) @ rd(sd["text_projection"].float())


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    return x / x.norm(dim=-1, keepdim=True)


def param_count(sd: SD) -> int:
    return sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k)
