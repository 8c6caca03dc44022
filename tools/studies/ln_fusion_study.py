#!/usr/bin/env python3
"""Error study for folding LayerNorm into the consumer GEMM, CPU only, before any kernel exists.

    LN(x) W^T + b  =  rstd * ( x (gamma o W)^T  -  mean * s )  +  c,     s_n = sum_k (gamma_k W_nk),  c_n = sum_k beta_k W_nk + b_n

Policies compared on the true-dimension ViT-B/32 image tower (hash-seeded weights, synthetic images), logits against N
random unit-norm class rows, all against the fp32 oracle:
  A  today's HIP path:   h = f16(LN_fp32(x)),  h . f16(W)^T  (fp32 accumulate)
  B  algebraic:          f16(x) . f16(gamma o W)^T (fp32 accumulate), row statistics from the fp32 stream (sum / sum of
                         squares), s from the ROUNDED folded weight (so the mean term cancels what the MFMA summed), c in fp32
Everything else (attention operands, fp32 residual, QuickGELU, c_proj) is rounded as the HIP path rounds it.
Also reported: |mean| / std of the residual rows (the cancellation risk of B) and max |x| (f16 range)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from hgr_net_amd import synth
from oracle import clip_ref

rd = clip_ref.round_f16 if (len(sys.argv) < 2 or sys.argv[1] == "f16") else clip_ref.round_bf16
torch.manual_seed(0)
cfg_name = "ViT-B/32"
sd = synth.clip_state_dict(cfg_name, 0)
B = 8
img = synth.images(B, 224, 4321)
stats = {"mean_over_std": 0.0, "max_abs_x": 0.0}


def ln_linear(x, sd, ln, wkey, bkey, policy):
    w, b = sd[wkey].float(), sd[bkey].float()
    g, be = sd[ln + ".weight"].float(), sd[ln + ".bias"].float()
    if policy == "ref":
        return clip_ref.layer_norm(x, sd, ln) @ w.t() + b
    if policy == "A":
        return rd(clip_ref.layer_norm(x, sd, ln)) @ rd(w).t() + b
    mean = x.mean(-1, keepdim=True)
    var = (x * x).mean(-1, keepdim=True) - mean * mean            # sum / sum-of-squares form, fp32
    rstd = torch.rsqrt(var + 1e-5)
    stats["mean_over_std"] = max(stats["mean_over_std"], float((mean.abs() * rstd).max()))
    stats["max_abs_x"] = max(stats["max_abs_x"], float(x.abs().max()))
    wf = rd(w * g)                                                 # folded weight, rounded once
    s = wf.sum(-1)
    c = w @ be + b
    acc = rd(x) @ wf.t()
    return rstd * (acc - mean * s) + c


def block(x, sd, p, heads, policy):
    b, l, wd = x.shape
    d = wd // heads
    qkv = ln_linear(x, sd, p + ".ln_1", p + ".attn.in_proj_weight", p + ".attn.in_proj_bias", policy)
    if policy != "ref":
        qkv = rd(qkv)
    q, k, v = qkv.split(wd, dim=-1)
    q = q.reshape(b, l, heads, d).transpose(1, 2); k = k.reshape(b, l, heads, d).transpose(1, 2); v = v.reshape(b, l, heads, d).transpose(1, 2)
    pm = torch.softmax((q @ k.transpose(-1, -2)) * d ** -0.5, dim=-1)
    o = ((rd(pm) if policy != "ref" else pm) @ v).transpose(1, 2).reshape(b, l, wd)
    r = (lambda t: t) if policy == "ref" else rd
    x = x + r(o) @ r(sd[p + ".attn.out_proj.weight"].float()).t() + sd[p + ".attn.out_proj.bias"].float()
    h = ln_linear(x, sd, p + ".ln_2", p + ".mlp.c_fc.weight", p + ".mlp.c_fc.bias", policy)
    u = clip_ref.quick_gelu(h)
    return x + r(u) @ r(sd[p + ".mlp.c_proj.weight"].float()).t() + sd[p + ".mlp.c_proj.bias"].float()


def tower(policy):
    w = sd["visual.conv1.weight"].float()
    width, _, ps, _ = w.shape
    g = 224 // ps
    r = (lambda t: t) if policy == "ref" else rd
    patches = img.float().reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps)
    x = r(patches) @ r(w.reshape(width, -1)).t()
    x = torch.cat([sd["visual.class_embedding"].float().expand(B, 1, width), x], dim=1) + sd["visual.positional_embedding"].float()
    x = clip_ref.layer_norm(x, sd, "visual.ln_pre")
    for i in range(12):
        x = block(x, sd, f"visual.transformer.resblocks.{i}", width // 64, policy)
    x = clip_ref.layer_norm(x[:, 0, :], sd, "visual.ln_post")
    f = r(x) @ r(sd["visual.proj"].float())
    return f / f.norm(dim=-1, keepdim=True)


with torch.no_grad():
    z = torch.randn(21841, 512)
    z = z / z.norm(dim=1, keepdim=True)
    fr = tower("ref")
    lr = fr @ z.t()
    for pol in ("A", "B"):
        f = tower(pol)
        lg = (rd(f) @ rd(z).t())
        print(f"policy {pol}: max |logit - fp32| = {float((lg - lr).abs().max()):.3e}   max |feature - fp32| = {float((f - fr).abs().max()):.3e}")
    print("residual rows: max |mean| / std =", round(stats["mean_over_std"], 3), "  max |x| =", round(stats["max_abs_x"], 2))
