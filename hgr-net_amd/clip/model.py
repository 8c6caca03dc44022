"""CLIP with the reference's module tree / ``state_dict`` schema and a libhgr.so forward.

The module tree below exists to keep checkpoints interchangeable with the reference
(clip/model.py:239-368; key dump in SURVEY.md section 8b): stock ``nn.LayerNorm`` / ``nn.Linear`` /
``nn.MultiheadAttention`` / ``nn.Conv2d`` objects are used purely as *parameter containers*, so that
``load_state_dict`` of a reference checkpoint, ``named_parameters()`` and helper functions that walk
the tree with ``isinstance`` checks (utils.py:98-123) behave the same.  None of their ``forward``
methods is ever called: ``encode_image`` / ``encode_text`` run the hand-written gfx950 kernels through
the C ABI (``hgr_net_amd.ops``) on 16-bit copies of the weights, with an fp32 residual stream.

Token layout is batch-major ([B*L, W] rows = b*L + t) instead of the reference's sequence-first
[L, B, W]; it is the same arithmetic and keeps every per-token row contiguous.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import (EPI_BIAS, EPI_BIAS_ADD16_RELU, EPI_BIAS_QUICKGELU, EPI_BIAS_RELU, EPI_BIAS_RESIDUAL, EPI_NONE,
                    HgrError)


# ------------------------------------------------------------------------------------------------
# parameter containers (schema only)
# ------------------------------------------------------------------------------------------------
class QuickGELU(nn.Module):
    """Placeholder so ``mlp`` keeps the reference's child names (c_fc, gelu, c_proj); the activation
    itself is fused into the c_fc GEMM epilogue (HGR_EPI_BIAS_QUICKGELU)."""


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model: int, n_head: int):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model)


class Transformer(nn.Module):
    def __init__(self, width: int, layers: int, heads: int):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        grid = input_resolution // patch_size
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(grid * grid + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)),
                                                         ("0", nn.Conv2d(inplanes, planes * 4, 1, bias=False)),
                                                         ("1", nn.BatchNorm2d(planes * 4))]))


class _AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.num_heads = num_heads


class ModifiedResNet(nn.Module):
    """Schema container for the RN towers (clip/model.py:93-150); the forward is `_rn_forward` below."""

    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim, self.input_resolution = output_dim, input_resolution
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        inplanes = width
        for li, nb in enumerate(layers):
            planes = width * 2 ** li
            blocks = [_Bottleneck(inplanes, planes, 1 if li == 0 else 2)]
            inplanes = planes * 4
            blocks += [_Bottleneck(inplanes, planes) for _ in range(1, nb)]
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.attnpool = _AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)


# ------------------------------------------------------------------------------------------------
# weight preparation and the transformer engine
# ------------------------------------------------------------------------------------------------
def _w16(t: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    return t.detach().to(dt).contiguous()


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


class _Block16:
    """16-bit GEMM weights + fp32 vectors of one residual block."""

    def __init__(self, blk: ResidualAttentionBlock, dt: torch.dtype):
        self.ln1 = (_f32(blk.ln_1.weight), _f32(blk.ln_1.bias))
        self.ln2 = (_f32(blk.ln_2.weight), _f32(blk.ln_2.bias))
        self.w_in, self.b_in = _w16(blk.attn.in_proj_weight, dt), _f32(blk.attn.in_proj_bias)
        self.w_out, self.b_out = _w16(blk.attn.out_proj.weight, dt), _f32(blk.attn.out_proj.bias)
        self.w_fc, self.b_fc = _w16(blk.mlp.c_fc.weight, dt), _f32(blk.mlp.c_fc.bias)
        self.w_proj, self.b_proj = _w16(blk.mlp.c_proj.weight, dt), _f32(blk.mlp.c_proj.bias)
        # LayerNorm folded into its consumer GEMM (hgr_gemm_nt_ln): W' = gamma o W rounded ONCE to the MFMA type,
        # s_n = sum_k W'_nk over the ROUNDED values (it must cancel exactly what the MFMA summed), c = W beta + b in fp32
        self.wf_in, self.s_in, self.c_in = self._fold(blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.ln_1, dt)
        self.wf_fc, self.s_fc, self.c_fc = self._fold(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, blk.ln_2, dt)
        self.eps1, self.eps2 = float(blk.ln_1.eps), float(blk.ln_2.eps)

    @staticmethod
    def _fold(w, b, ln, dt):
        w32, g32, be32 = w.detach().float(), ln.weight.detach().float(), ln.bias.detach().float()
        wf = (w32 * g32[None, :]).to(dt).contiguous()
        return wf, wf.float().sum(dim=1).contiguous(), (w32 @ be32 + b.detach().float()).contiguous()


class _Workspace:
    """Named device buffers reused across calls (keeps the hot loop free of allocator traffic)."""

    def __init__(self):
        self._b: Dict[str, torch.Tensor] = {}
        self.epoch = 0          # bumped whenever a buffer is (re)allocated: captured HIP graphs hold the old pointers

    def const(self, name: str, build) -> torch.Tensor:
        """A small constant device tensor built once (index vectors): never re-created inside a graph capture."""
        t = self._b.get("const." + name)
        if t is None:
            t = self._b["const." + name] = build()
            self.epoch += 1
        return t

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        n = int(np.prod(shape))
        t = self._b.get(name)
        if t is None or t.numel() < n or t.dtype != dtype or t.device != device:
            t = torch.empty(n, dtype=dtype, device=device)
            self._b[name] = t
            self.epoch += 1
        return t[:n].view(*shape)


# ViT image batches of at least IMG_STREAMS_MIN_ROWS patch rows run as HGR_IMG_STREAMS independent slices on as many streams
IMG_STREAMS = max(1, int(os.environ.get("HGR_IMG_STREAMS", "1")))
IMG_STREAMS_MIN_ROWS = 8192
CLS_LAST = os.environ.get("HGR_CLS_LAST", "1") != "0"      # HGR_CLS_LAST=0: the last image block runs out_proj / MLP on every token (A/B runs, tests)
LN_FUSED = os.environ.get("HGR_LN_FUSED", "1") != "0"     # HGR_LN_FUSED=0: separate LayerNorm launches (the first build's path), for A/B runs
QKV_MHA = os.environ.get("HGR_QKV_MHA", "1") != "0"       # HGR_QKV_MHA=0: the in_proj GEMM writes qkv and hgr_mha reads it back (two launches)


def ln_fusable(w: int, m: int = 0) -> bool:
    """The LayerNorm-folded GEMM pair needs the row width to be a multiple of 128 (two 64-column statistic slots per load) and
    32-bit operand offsets: its widest A operand is u16 [m, 4w] in 16 bit, which must stay below 4 GB (ViT-L/14@336 above ~900
    images does not: those batches take the unfused LayerNorm + hgr_gemm_nt route, whose 128 / 256 tile kernels use 64-bit rows)."""
    return LN_FUSED and w % 128 == 0 and m * 4 * w * 2 < 2 ** 32


def _run_blocks(x: Optional[torch.Tensor], blocks, heads: int, b: int, l: int, causal: bool, dt: torch.dtype, ws: _Workspace, tag: str,
                taps: Optional[dict] = None, tap_prefix: str = "", pair=None, stats: Optional[torch.Tensor] = None,
                flag: Optional[torch.Tensor] = None, cls_only_last: bool = False):
    """The residual stack (clip/model.py:185-188 per block).

    Fused form (``pair`` = (xh, xl): the residual stream as a 16-bit pair, x = xh + xl, with its LayerNorm slot statistics in
    ``stats`` - from hgr_vit_embed_ln_stats / hgr_row_stats16): the LayerNorms are folded into the GEMMs around them, 5 launches
    per block and no LayerNorm pass:
        QKV = LN-folded GEMM(xh) -> attention [one launch, no qkv tensor, for sequences of <= 64 tokens: hgr_gemm_nt_ln_mha]
        -> (xh, xl) += out GEMM (+ stats) -> u = LN-folded GEMM(xh, QuickGELU) -> (xh, xl) += proj GEMM (+ stats)
    ``flag``: the producers' range guard (ops.gemm_nt_res_stats), see CLIP._ln_check.
    ``cls_only_last`` (image tower): the visual head reads ONLY the class token of the last block's output
    (``x = self.ln_post(x[:, 0, :])``, clip/model.py:231), and out_proj / ln_2 / the MLP act on every token independently (:186-187), so
    the last block runs them on the b class-token rows alone - strided views into the same pair, same kernels, the same bits for
    those rows; its attention still sees every key and value.  The other rows of (xh, xl) keep the previous block's values.
    Otherwise, on the fp32 stream x [b*l, w]: LN -> QKV GEMM(+bias) -> attention -> out GEMM(+bias, +residual) -> LN -> fc
    GEMM(+bias, QuickGELU) -> proj GEMM(+bias, +residual): 7 launches per block."""
    m, w = (pair[0] if pair is not None else x).shape
    dev = (pair[0] if pair is not None else x).device
    qkv = ws.get(tag + ".qkv", (m, 3 * w), dt, dev)
    att = ws.get(tag + ".att", (m, w), dt, dev)
    u16 = ws.get(tag + ".u16", (m, 4 * w), dt, dev)
    if pair is not None:
        xh, xl = pair
        for i, k in enumerate(blocks):
            last_cls = cls_only_last and i == len(blocks) - 1 and l > 1 and 3 * w * l < (1 << 23)    # hgr_gemm_nt_ln: ldc < 2^23
            fused_attn = QKV_MHA and not last_cls and ops.ln_mha_ok(w, l, m, xh.stride(0))
            if last_cls:
                # keys and values of every token, queries of the class tokens only (the one attention row that is read)
                ch, cl = xh.view(b, l, w)[:, 0, :], xl.view(b, l, w)[:, 0, :]                  # row stride l * w
                ops.gemm_nt_ln(xh, k.wf_in[w:], qkv[:, w:], k.s_in[w:], k.c_in[w:], stats, k.eps1, tag="kv")
                ops.split_point("class_token_tail")      # everything behind this line works on b rows: a few dozen workgroups per launch
                st_q = ws.get(tag + ".stats_q", (b, w // 64, 2), torch.float32, dev)
                idx = ws.const(f"cls_rows.{b}.{l}", lambda: (torch.arange(b, dtype=torch.int32, device=dev) * l).contiguous())
                ops.rows_gather(stats.view(b * l, -1), idx, st_q.view(b, -1))                   # the class rows' ln_1 statistics, compact
                ops.gemm_nt_ln(ch, k.wf_in[:w], qkv.view(b, l, 3 * w)[:, 0, :w], k.s_in[:w], k.c_in[:w], st_q, k.eps1, tag="q_cls")
            elif fused_attn:
                ops.gemm_nt_ln_mha(xh, k.wf_in, att, k.s_in, k.c_in, stats, b, l, heads, causal, k.eps1)    # no qkv tensor, no attention launch
            else:
                ops.gemm_nt_ln(xh, k.wf_in, qkv, k.s_in, k.c_in, stats, k.eps1, tag="qkv")
            if not fused_attn:
                ops.mha(qkv, att, b, l, heads, causal, q_rows=1 if last_cls else 0)
            if last_cls:
                ca = att.view(b, l, w)[:, 0, :]
                st_c = ws.get(tag + ".stats_cls", (b, w // 64, 2), torch.float32, dev)
                u_c = ws.get(tag + ".u16_cls", (b, 4 * w), dt, dev)
                ops.gemm_nt_res_stats(ca, k.w_out, ch, cl, k.b_out, st_c, tag="out_cls", flag=flag)
                ops.gemm_nt_ln(ch, k.wf_fc, u_c, k.s_fc, k.c_fc, st_c, k.eps2, quickgelu=True, tag="fc_cls")
                ops.gemm_nt_res_stats(u_c, k.w_proj, ch, cl, k.b_proj, st_c, tag="proj_cls", flag=flag)
                break
            ops.gemm_nt_res_stats(att, k.w_out, xh, xl, k.b_out, stats, tag="out", flag=flag)
            ops.gemm_nt_ln(xh, k.wf_fc, u16, k.s_fc, k.c_fc, stats, k.eps2, quickgelu=True, tag="fc")
            ops.gemm_nt_res_stats(u16, k.w_proj, xh, xl, k.b_proj, stats, tag="proj", flag=flag)
            if taps is not None:
                taps[f"{tap_prefix}.resblocks.{i}"] = ops.pair_value(xh, xl).view(b, l, w)
        return None
    h16 = ws.get(tag + ".h16", (m, w), dt, dev)
    for i, k in enumerate(blocks):
        ops.layernorm(x, k.ln1[0], k.ln1[1], h16)
        ops.gemm_nt(h16, k.w_in, qkv, bias=k.b_in, epilogue=EPI_BIAS)
        ops.mha(qkv, att, b, l, heads, causal)
        ops.gemm_nt(att, k.w_out, x, bias=k.b_out, residual=x, epilogue=EPI_BIAS_RESIDUAL)
        ops.layernorm(x, k.ln2[0], k.ln2[1], h16)
        ops.gemm_nt(h16, k.w_fc, u16, bias=k.b_fc, epilogue=EPI_BIAS_QUICKGELU)
        ops.gemm_nt(u16, k.w_proj, x, bias=k.b_proj, residual=x, epilogue=EPI_BIAS_RESIDUAL)
        if taps is not None:
            taps[f"{tap_prefix}.resblocks.{i}"] = x.view(b, l, w).clone()
    return x



# ------------------------------------------------------------------------------------------------
# ModifiedResNet engine: NHWC 16-bit activations, BN folded, 1x1 convs = GEMMs, 3x3 = implicit GEMM
# ------------------------------------------------------------------------------------------------
def _cpad(c: int) -> int:
    """Channel counts that become a GEMM K (block inputs / planes) are stored padded to a multiple of 64 with zero
    weights on both sides: RN50/RN101 (64, 128, ...) are untouched, RN50x4 stores 80 -> 128 and 160 -> 192, RN50x16
    96 -> 128.  The padded activations are exact zeros (zero weight rows, zero bias, ReLU), so results do not change."""
    return (c + 63) // 64 * 64


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, dt: torch.dtype, cin_p: Optional[int] = None, cout_p: Optional[int] = None):
    """Inference BatchNorm folded into the preceding conv (legal: the model is always in eval(),
    model/clip_tree.py:46).  Returns (w16 [cout_p, Kp] in (ky, kx, c) order with c padded to cin_p, bias fp32 [cout_p])."""
    w = conv.weight.detach().float()
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    b = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    w = (w * s.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)                                   # [Cout, kh, kw, Cin]
    cout, kh, kw, cin = w.shape
    cin_p, cout_p = cin_p or cin, cout_p or cout
    k = kh * kw * cin_p
    kp = (k + 63) // 64 * 64
    out = torch.zeros(cout_p, kp, dtype=dt, device=w.device)
    out[:cout, :k].view(cout, kh, kw, cin_p)[..., :cin] = w.to(dt)
    bias = torch.zeros(cout_p, dtype=torch.float32, device=w.device)
    bias[:cout] = b
    return out.contiguous(), bias.contiguous()


class _RNBlock16:
    def __init__(self, blk, dt):
        cin, pl = blk.conv1.weight.shape[1], blk.conv1.weight.shape[0]
        cin_p, pl_p = _cpad(cin), _cpad(pl)
        self.c1 = _fold(blk.conv1, blk.bn1, dt, cin_p, pl_p)
        self.c2 = _fold(blk.conv2, blk.bn2, dt, pl_p, pl_p)
        self.c3 = _fold(blk.conv3, blk.bn3, dt, pl_p, None)
        self.down = _fold(blk.downsample[1], blk.downsample[2], dt, cin_p, None) if blk.downsample is not None else None
        self.stride = blk.stride
        self.planes = pl_p                     # stored (padded) width of the block's inner activations
        self.out = blk.conv3.weight.shape[0]   # 4 * planes, always a multiple of 64


def _rn_prepare(v: "ModifiedResNet", dt: torch.dtype) -> dict:
    width = v.conv3.weight.shape[0]
    if width % 16:
        raise NotImplementedError(f"RN width {width}: channel counts must be multiples of 8 (the stem runs at width / 2)")
    p = {"stem": [_fold(v.conv1, v.bn1, dt), _fold(v.conv2, v.bn2, dt), _fold(v.conv3, v.bn3, dt, None, _cpad(width))]}
    p["blocks"] = [_RNBlock16(b, dt) for li in (1, 2, 3, 4) for b in getattr(v, f"layer{li}")]
    a = v.attnpool
    p["pos"] = _f32(a.positional_embedding)
    for nm in ("q_proj", "k_proj", "v_proj", "c_proj"):
        lin = getattr(a, nm)
        p[nm] = (_w16(lin.weight, dt), _f32(lin.bias))
    return p


def _rn_forward(v: "ModifiedResNet", p: dict, image: torch.Tensor, dt: torch.dtype, ws: _Workspace) -> torch.Tensor:
    """ModifiedResNet.forward (clip/model.py:135-150) on libhgr: stem (3 conv+BN+ReLU, avgpool) ->
    4 stages of Bottlenecks (clip/model.py:40-53) -> AttentionPool2d (clip/model.py:66-90)."""
    dev = image.device
    b, _, r, _ = image.shape
    width = p["stem"][2][0].shape[0]                      # stored width (padded to 64, _cpad)
    h = (r - 1) // 2 + 1
    m = b * h * h
    (w1, b1), (w2, b2), (w3, b3) = p["stem"]
    a1 = ws.get("r.a", (m, w1.shape[0]), dt, dev)
    if r % 4 == 0 and w1.shape[0] <= 48:
        ops.stem_conv1(image, w1, b1, a1)                 # conv + BN + ReLU straight from the fp32 image
    else:
        col = ws.get("r.col", (m, 64), dt, dev)
        ops.stem_im2col(image, col)
        ops.gemm_nt(col, w1, a1, bias=b1, epilogue=EPI_BIAS_RELU)
    a2 = ws.get("r.b", (m, w2.shape[0]), dt, dev)
    ops.conv3x3_nhwc(a1, w2, b2, a2, b, h, h, w1.shape[0])
    if w2.shape[0] == 32 and width == 64 and h % 2 == 0:      # RN50 / RN101: conv3 + BN + ReLU + AvgPool2d(2) in one kernel
        h //= 2
        x = ws.get("r.x0", (b * h * h, width), dt, dev)
        ops.conv3x3_pool2_nhwc(a2, w3, b3, x, b, 2 * h, 2 * h, 32)
    else:
        a3 = ws.get("r.c", (m, w3.shape[0]), dt, dev)
        ops.conv3x3_nhwc(a2, w3, b3, a3, b, h, h, w2.shape[0])
        h //= 2
        x = ws.get("r.x0", (b * h * h, width), dt, dev)
        ops.avgpool2_nhwc(a3, x, b, 2 * h, 2 * h, width)
    cin, flip = width, 0
    for k in p["blocks"]:
        m = b * h * h
        pl = k.planes
        t1 = ws.get("r.t1", (m, pl), dt, dev)
        ops.gemm_nt(x, k.c1[0], t1, bias=k.c1[1], epilogue=EPI_BIAS_RELU)
        t2 = ws.get("r.t2", (m, pl), dt, dev)
        ops.conv3x3_nhwc(t1, k.c2[0], k.c2[1], t2, b, h, h, pl)
        ho = h // k.stride
        mo = b * ho * ho
        xin = x
        if k.stride > 1:                                   # anti-aliased stride: avgpool after conv2 and on the skip path
            t2p = ws.get("r.t2p", (mo, pl), dt, dev)
            ops.avgpool2_nhwc(t2, t2p, b, h, h, pl)
            t2 = t2p
            if k.down is not None:
                xin = ws.get("r.xp", (mo, cin), dt, dev)
                ops.avgpool2_nhwc(x, xin, b, h, h, cin)
        idn = xin
        if k.down is not None:
            idn = ws.get("r.idn", (mo, k.out), dt, dev)
            ops.gemm_nt(xin, k.down[0], idn, bias=k.down[1], epilogue=EPI_BIAS)
        out = ws.get(f"r.out{flip}", (mo, k.out), dt, dev)
        ops.gemm_nt(t2, k.c3[0], out, bias=k.c3[1], residual=idn, epilogue=EPI_BIAS_ADD16_RELU)
        x, cin, h, flip = out, k.out, ho, flip ^ 1
    # attention pool: only token 0's query is ever used (the module returns x[0])
    a = v.attnpool
    e, l = cin, h * h + 1
    tok = ws.get("r.tok", (b * l, e), dt, dev)
    ops.attnpool_tokens(x, p["pos"], tok, b, h, e)
    kk = ws.get("r.k", (b * l, e), dt, dev)
    vv = ws.get("r.v", (b * l, e), dt, dev)
    ops.gemm_nt(tok, p["k_proj"][0], kk, bias=p["k_proj"][1], epilogue=EPI_BIAS)
    ops.gemm_nt(tok, p["v_proj"][0], vv, bias=p["v_proj"][1], epilogue=EPI_BIAS)
    q = ws.get("r.q", (b, e), torch.float32, dev)
    ops.gemm_nt(tok.view(b, l * e)[:, :e], p["q_proj"][0], q, bias=p["q_proj"][1], epilogue=EPI_BIAS)   # rows b*l (token 0)
    o16 = ws.get("r.o", (b, e), dt, dev)
    ops.attnpool_attend(q, kk, vv, o16, b, l, a.num_heads)
    out = torch.empty((b, v.output_dim), dtype=torch.float32, device=dev)
    ops.gemm_nt(o16, p["c_proj"][0], out, bias=p["c_proj"][1], epilogue=EPI_BIAS)
    return out

# ------------------------------------------------------------------------------------------------
# CLIP
# ------------------------------------------------------------------------------------------------
class CLIP(nn.Module):
    """Same constructor signature as the reference CLIP (clip/model.py:240-254) plus the two compute
    dtypes.  ``image_dtype`` / ``text_dtype``: 16-bit MFMA input type of each tower ("f16" | "bf16");
    accumulation, LayerNorm, softmax and the residual stream are fp32 in both.

    Default f16: it is the reference's own GPU dtype (clip/model.py:371-392), published checkpoints are
    fp16 so the weights enter the MFMA exactly, it runs at the same MFMA rate as bf16 on gfx950, and it
    keeps the logits within 1e-3 of the fp32 reference with ~8x margin (measured max 1.2e-4 vs 1.2e-3
    for bf16 on ViT-B/32, DESIGN.md "Precision").  bf16 is kept for ranges fp16 cannot hold."""

    def __init__(self, embed_dim: int, image_resolution: int, vision_layers: Union[Tuple[int, int, int, int], int],
                 vision_width: int, vision_patch_size: int, context_length: int, vocab_size: int,
                 transformer_width: int, transformer_heads: int, transformer_layers: int,
                 image_dtype: str = "f16", text_dtype: str = "f16"):
        super().__init__()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            self.visual = ModifiedResNet(vision_layers, embed_dim, vision_width * 32 // 64, image_resolution, vision_width)
        else:
            self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                            vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width).normal_(std=0.01))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim).normal_(std=transformer_width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.image_dtype = ops.TORCH16[ops.dtype_code(image_dtype)]
        self.text_dtype = ops.TORCH16[ops.dtype_code(text_dtype)]
        self.text_chunk = 8192          # prompts per text-tower pass (bounds the workspace; result is chunk-invariant)
        self._prep: dict = {}
        self._ws = _Workspace()
        # LayerNorm folding keeps the residual stream's high half in 16 bit; the reference keeps LayerNorm in fp32 because trained
        # checkpoints carry outlier channels (clip/model.py:153-159).  Range guard: every producer GEMM reports a 64-column slot whose
        # sum of squares leaves the safe range (or is inf / NaN) into _ln_flags[tower]; a tower whose flag trips runs unfused from
        # then on (fp32 stream + separate LayerNorm launches = HGR_LN_FUSED=0 for that tower).
        self._ln_flags: Optional[torch.Tensor] = None
        self._ln_off: set = set()
        self._ln_checked: dict = {}
        self._img_tag = "v"

    # -- LayerNorm-folding range guard ------------------------------------------------------------
    def _ln_flag(self, tower: str, dev) -> torch.Tensor:
        if self._ln_flags is None or self._ln_flags.device != torch.device(dev):
            self._ln_flags = torch.zeros(2, dtype=torch.int32, device=dev)
        i = 0 if tower == "v" else 1
        return self._ln_flags[i:i + 1]

    def ln_guard_tripped(self) -> dict:
        """{tower: largest offending slot sum of squares (inf / nan included)} of the towers whose folded residual stream left the
        guarded range since the flags were last cleared; empty = in range.  One small D2H copy (synchronises)."""
        if self._ln_flags is None:
            return {}
        bits = self._ln_flags.cpu().numpy().view(np.uint32)
        vals = bits.view(np.float32)
        return {t: float(vals[i]) for i, t in enumerate(("image", "text")) if bits[i]}

    def _ln_check(self, tower: str, dev) -> bool:
        """After the FIRST folded pass of a tower under the current weights (never inside a graph capture): read its flag; when it
        tripped, switch the tower to the unfused path for good, say so, and tell the caller to recompute.  Later passes are not
        checked in the hot loop (no synchronisation there); Evaluator.summary() reads the flags once more at the end."""
        fp = self._prep.get("fp")
        if self._ln_checked.get(tower) == fp or torch.cuda.is_current_stream_capturing():
            return False
        self._ln_checked[tower] = fp
        i = 0 if tower == "v" else 1
        bits = int(self._ln_flags[i].item()) & 0xFFFFFFFF
        if not bits:
            return False
        worst = float(np.array([bits], dtype=np.uint32).view(np.float32)[0])
        import warnings
        warnings.warn(f"hgr_net_amd: the {'image' if tower == 'v' else 'text'} tower's residual stream left the range the 16-bit LayerNorm-folded path "
                      f"is guarded for (a 64-column slot with sum of squares {worst:g} > {ops.LN_GUARD_SUMSQ:g}): this tower now runs with the fp32 "
                      f"stream and separate LayerNorm launches (the HGR_LN_FUSED=0 path).")
        self._ln_off.add(tower)
        self._ln_flags[i] = 0
        return True

    LN_POLL_EVERY = 64

    def poll_ln_guard(self, force: bool = False) -> bool:
        """The first-pass check (_ln_check) cannot see a LATER batch leave the guarded range.  Every LN_POLL_EVERY-th call (encode_image /
        encode_text outside a graph capture, and every replayed forward of tree_model) reads the flags once (one small D2H copy); a
        tripped tower is switched to the unfused path for good, with a warning: the batches since the previous poll may hold inf / NaN
        features, which the warning says.  Returns whether a tower was switched (callers holding HIP graphs must drop them)."""
        self._ln_polls = getattr(self, "_ln_polls", 0) + 1
        if self._ln_flags is None or torch.cuda.is_current_stream_capturing() or not (force or self._ln_polls % self.LN_POLL_EVERY == 0):
            return False
        tripped = self.ln_guard_tripped()
        if not tripped:
            return False
        import warnings
        for name, worst in tripped.items():
            t = "v" if name == "image" else "t"
            self._ln_off.add(t)
            warnings.warn(f"hgr_net_amd: the {name} tower's residual stream left the range the 16-bit LayerNorm-folded path is guarded for AFTER its "
                          f"first-pass check (slot sum of squares {worst:g} > {ops.LN_GUARD_SUMSQ:g}); up to {self.LN_POLL_EVERY} batches since the previous "
                          f"poll may hold inf / NaN features.  This tower now runs with the fp32 stream and separate LayerNorm launches.")
        self._ln_flags.zero_()
        return True

    # -- reference surface ---------------------------------------------------------------------
    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def _fingerprint(self):
        """(storage, torch version counter) of every parameter + ops.WEIGHTS_GEN: the fused optimizer writes the flat
        master buffer through a raw pointer, which moves neither data_ptr nor _version."""
        return (ops.WEIGHTS_GEN[0],) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _prepared(self) -> dict:
        """16-bit weight copies, rebuilt whenever a parameter was replaced or written in place
        (load_state_dict, .to(), a torch optimizer) or a libhgr optimizer step ran (ops.WEIGHTS_GEN)."""
        fp = self._fingerprint()
        if self._prep.get("fp") != fp:
            p: dict = {"fp": fp}
            # new weights: a tower that left the guarded range under the old ones gets the folded path back (and is checked again)
            self._ln_off.clear()
            self._ln_checked.clear()
            v = self.visual
            if isinstance(v, VisionTransformer):
                dt = self.image_dtype
                k = 3 * v.patch_size * v.patch_size
                kp = (k + 63) // 64 * 64
                wc = torch.zeros(v.conv1.weight.shape[0], kp, dtype=dt, device=v.conv1.weight.device)
                wc[:, :k] = v.conv1.weight.detach().reshape(v.conv1.weight.shape[0], -1).to(dt)
                p["conv_w"], p["kp"] = wc, kp
                # the same weight in (py, px, c) K order for the uint8 NHWC input path (hgr_im2col_patches_u8)
                p["conv_w_nhwc"] = v.conv1.weight.detach().permute(0, 2, 3, 1).reshape(v.conv1.weight.shape[0], -1).to(dt).contiguous()
                p["cls"], p["pos"] = _f32(v.class_embedding), _f32(v.positional_embedding)
                p["ln_pre"] = (_f32(v.ln_pre.weight), _f32(v.ln_pre.bias))
                p["ln_post"] = (_f32(v.ln_post.weight), _f32(v.ln_post.bias))
                p["proj_t"] = _w16(v.proj.detach().t(), dt)                     # [D, W] for the NT GEMM
                p["vblocks"] = [_Block16(b, dt) for b in v.transformer.resblocks]
            else:
                p["rn"] = _rn_prepare(v, self.image_dtype)
            dt = self.text_dtype
            p["tok"], p["tpos"] = _f32(self.token_embedding.weight), _f32(self.positional_embedding)
            p["ln_final"] = (_f32(self.ln_final.weight), _f32(self.ln_final.bias))
            p["tproj_t"] = _w16(self.text_projection.detach().t(), dt)
            p["tblocks"] = [_Block16(b, dt) for b in self.transformer.resblocks]
            self._prep = p
        return self._prep

    @torch.no_grad()
    def encode_image(self, image: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """fp32 [B, embed_dim] image features (clip/model.py:336 -> VisionTransformer.forward :219-236)."""
        v = self.visual
        if not image.is_cuda:
            raise HgrError("encode_image needs a device tensor: the product path has no CPU fallback")
        p = self._prepared()
        dt, ws, dev = self.image_dtype, self._ws, image.device
        u8 = image.dtype == torch.uint8                       # [B, R, R, 3] crops straight from the decoder: normalised in the patch kernel
        if u8:
            if not isinstance(v, VisionTransformer) or image.dim() != 4 or image.shape[-1] != 3 or (3 * v.patch_size ** 2) % 64:
                raise NotImplementedError("uint8 NHWC input is implemented for ViT towers with 3*P*P % 64 == 0")
            image = image.contiguous()
            b, r = image.shape[0], image.shape[1]
        else:
            image = image.float().contiguous()
            b, _, r, _ = image.shape
        if r != v.input_resolution:
            raise ValueError(f"expected {v.input_resolution}x{v.input_resolution} input, got {r}")
        if not isinstance(v, VisionTransformer):
            return _rn_forward(v, p["rn"], image, dt, ws)
        self.poll_ln_guard()
        out = torch.empty((b, v.output_dim), dtype=torch.float32, device=dev)
        ns = IMG_STREAMS if (taps is None and b >= 2 * IMG_STREAMS and b * (r // v.patch_size) ** 2 >= IMG_STREAMS_MIN_ROWS) else 1
        if ns == 1:
            tag = self._img_tag           # workspace set of this call ("v"; tree_model's pipelined evaluation alternates two sets)
            fused = self._vit_forward(image, p, out, tag, taps, u8)
            if fused and self._ln_check("v", dev):
                self._vit_forward(image, p, out, tag, taps, u8)          # the guard tripped: once more on the unfused path
            return out
        # The batch in ``ns`` independent slices on ``ns`` streams: the residual stack is a strict chain of launches, and a
        # launch whose tile count is not a multiple of the chip's workgroup slots ends in a partly empty round (N = 768 at
        # batch 512: 600 tiles on 512 slots).  With a second, independent chain in flight the other slice's workgroups take the
        # slots a launch's tail leaves idle.  Same kernels on the same rows: the bits do not depend on the split.
        main = torch.cuda.current_stream()
        side = self._side_streams(ns - 1, dev)
        cut = [b * i // ns for i in range(ns + 1)]
        fused = False
        for i in range(ns):
            st = main if i == 0 else side[i - 1]
            if i:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                fused |= self._vit_forward(image[cut[i]:cut[i + 1]], p, out[cut[i]:cut[i + 1]], f"v{i}" if i else "v", None, u8)
        for st in side:
            main.wait_stream(st)
        if fused and self._ln_check("v", dev):
            self._vit_forward(image, p, out, "v", None, u8)              # the guard tripped in some slice: the whole batch once more, unfused
        return out

    def _side_streams(self, n: int, dev):
        have = self.__dict__.setdefault("_streams", [])
        while len(have) < n:
            have.append(torch.cuda.Stream(device=dev))
        return have[:n]

    def _vit_forward(self, image: torch.Tensor, p: dict, out: torch.Tensor, tag: str, taps: Optional[dict], u8: bool) -> bool:
        """VisionTransformer.forward (clip/model.py:219-236) of one slice of the batch on the current stream, features into ``out``.
        Returns whether the LayerNorm-folded path ran."""
        v = self.visual
        dt, ws, dev = self.image_dtype, self._ws, image.device
        b, r = image.shape[0], v.input_resolution
        ps = v.patch_size
        g = r // ps
        gg, l, w = g * g, g * g + 1, v.conv1.weight.shape[0]
        patches = ws.get(tag + ".patches", (b * gg, p["kp"]), dt, dev)
        if u8:
            ops.im2col_patches_u8(image, patches, ps)
        else:
            ops.im2col_patches(image, patches, ps)
        pe = ws.get(tag + ".pe", (b * gg, w), torch.float32, dev)
        ops.gemm_nt(patches, p["conv_w_nhwc"] if u8 else p["conv_w"], pe, tag="patch")
        cls16 = ws.get(tag + ".cls16", (b, w), dt, dev)
        fused = ln_fusable(w, b * l) and "v" not in self._ln_off
        if fused:
            xh = ws.get(tag + ".xh", (b * l, w), dt, dev)
            xl = ws.get(tag + ".xl", (b * l, w), ops.PAIR_LO, dev)
            stats = ws.get(tag + ".stats", (b * l, w // 64, 2), torch.float32, dev)
            ops.vit_embed_ln_stats(pe, p["cls"], p["pos"], p["ln_pre"][0], p["ln_pre"][1], xh, xl, stats, b, gg)
            if taps is not None:
                taps["visual.ln_pre"] = ops.pair_value(xh, xl).view(b, l, w)
            _run_blocks(None, p["vblocks"], w // 64, b, l, False, dt, ws, tag, taps, "visual.transformer", (xh, xl), stats, self._ln_flag("v", dev),
                        cls_only_last=taps is None and CLS_LAST)
            if w <= 1920 and w % 32 == 0 and v.output_dim % 4 == 0:
                ops.vit_head(xh, xl, l, p["ln_post"][0], p["ln_post"][1], float(v.ln_post.eps), p["proj_t"], out)   # ln_post(class tokens) @ proj
                return fused
            cls32 = ws.get(tag + ".cls32", (b, w), torch.float32, dev)
            ops.pair_rows_f32(xh, xl, cls32, row_mul=l)                          # the class tokens back in fp32 for ln_post
            ops.layernorm(cls32, p["ln_post"][0], p["ln_post"][1], cls16, rows=b)
        else:
            x = ws.get(tag + ".x", (b * l, w), torch.float32, dev)
            ops.vit_embed_ln(pe, p["cls"], p["pos"], p["ln_pre"][0], p["ln_pre"][1], x, b, gg)
            if taps is not None:
                taps["visual.ln_pre"] = x.view(b, l, w).clone()
            _run_blocks(x, p["vblocks"], w // 64, b, l, False, dt, ws, tag, taps, "visual.transformer")
            ops.layernorm(x, p["ln_post"][0], p["ln_post"][1], cls16, rows=b, row_mul=l)
        ops.gemm_nt(cls16, p["proj_t"], out)
        return fused

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor, trim: bool = True, ctx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 [n, embed_dim] text features (clip/model.py:339-352).  With ``trim`` the blocks run on
        positions <= max(EOT) only: exact, because the mask is causal and only the EOT row is read.
        ``ctx`` [n_ctx, W]: CoOp learnable context spliced over token positions 1..n_ctx (model/CoOp.py:98-113)."""
        if not text.is_cuda:
            raise HgrError("encode_text needs a device tensor: the product path has no CPU fallback")
        p = self._prepared()
        dt, ws, dev = self.text_dtype, self._ws, text.device
        text = text.long()
        n, ctx_len = text.shape
        w = self.transformer.width
        d = self.text_projection.shape[1]
        out = torch.empty((n, d), dtype=torch.float32, device=dev)
        eot = torch.empty(n, dtype=torch.int32, device=dev)
        ops.eot_index(text, eot)
        l = int(eot.max().item()) + 1 if trim else ctx_len
        s = 0
        while s < n:
            e = min(n, s + self.text_chunk)
            c = e - s
            x = ws.get("t.x", (c * l, w), torch.float32, dev)
            ops.text_embed(text[s:e], p["tok"], p["tpos"], x, l)
            if ctx is not None:
                ops.ctx_splice(x, ctx.detach().float().contiguous(), p["tpos"], c, l)
            f16 = ws.get("t.f16", (c, w), dt, dev)
            fused = ln_fusable(w, c * l) and "t" not in self._ln_off
            if fused:
                xh = ws.get("t.xh", (c * l, w), dt, dev)
                xl = ws.get("t.xl", (c * l, w), ops.PAIR_LO, dev)
                stats = ws.get("t.stats", (c * l, w // 64, 2), torch.float32, dev)
                ops.row_stats16(x, xh, xl, stats)
                _run_blocks(None, p["tblocks"], w // 64, c, l, True, dt, ws, "t", pair=(xh, xl), stats=stats, flag=self._ln_flag("t", dev))
                if self._ln_check("t", dev):
                    continue                                             # the guard tripped: this chunk once more, unfused
                e32 = ws.get("t.e32", (c, w), torch.float32, dev)
                ops.pair_rows_f32(xh, xl, e32, row_mul=l, row_idx=eot[s:e])       # the EOT rows back in fp32 for ln_final
                ops.layernorm(e32, p["ln_final"][0], p["ln_final"][1], f16, rows=c)
            else:
                _run_blocks(x, p["tblocks"], w // 64, c, l, True, dt, ws, "t")
                ops.layernorm(x, p["ln_final"][0], p["ln_final"][1], f16, rows=c, row_mul=l, row_idx=eot[s:e])
            ops.gemm_nt(f16, p["tproj_t"], out[s:e])
            s = e
        return out

    def forward(self, image, text):
        """Scaled cosine logits (clip/model.py:354-368); L2 norm and the product run in libhgr."""
        fi, ft = self.encode_image(image), self.encode_text(text)
        dt = self.image_dtype
        fi16 = torch.empty(fi.shape, dtype=dt, device=fi.device)
        ft16 = torch.empty(ft.shape, dtype=dt, device=ft.device)
        ops.l2norm_rows(fi, y16=fi16)
        ops.l2norm_rows(ft, y16=ft16)
        lg = torch.empty((fi.shape[0], (ft.shape[0] + 3) // 4 * 4), dtype=torch.float32, device=fi.device)
        ops.gemm_nt(fi16, ft16, lg, n=ft.shape[0])
        lg = lg[:, : ft.shape[0]] * self.logit_scale.exp()
        return lg, lg.t()


def infer_config(state_dict: dict) -> dict:
    """Architecture from tensor shapes, the rule of the reference's ``build_model`` (clip/model.py:395-418)."""
    vit = "visual.proj" in state_dict
    if vit:
        vw = state_dict["visual.conv1.weight"].shape[0]
        vl = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        ps = state_dict["visual.conv1.weight"].shape[-1]
        res = ps * round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    else:
        vl = tuple(len({k.split(".")[2] for k in state_dict if k.startswith(f"visual.layer{b}")}) for b in (1, 2, 3, 4))
        vw = state_dict["visual.layer1.0.conv1.weight"].shape[0]
        ow = round((state_dict["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        assert ow ** 2 + 1 == state_dict["visual.attnpool.positional_embedding"].shape[0]
        ps, res = None, ow * 32
    wt = state_dict["ln_final.weight"].shape[0]
    return dict(embed_dim=state_dict["text_projection"].shape[1], image_resolution=res, vision_layers=vl,
                vision_width=vw, vision_patch_size=ps, context_length=state_dict["positional_embedding"].shape[0],
                vocab_size=state_dict["token_embedding.weight"].shape[0], transformer_width=wt, transformer_heads=wt // 64,
                transformer_layers=len({k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")}))


def build_model(state_dict: dict, image_dtype: str = "f16", text_dtype: str = "f16") -> CLIP:
    """Construct from a reference-format state_dict (clip/model.py:395-432).  Master parameters stay
    fp32 (the reference casts them to fp16 here; the 16-bit copies the kernels read are made lazily)."""
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    model = CLIP(**infer_config(sd), image_dtype=image_dtype, text_dtype=text_dtype)
    model.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()})
    return model.eval()
