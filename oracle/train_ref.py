"""ORACLE (test infrastructure, never a product path): CPU fp32 restatement of one training step of the
reference - ``tree_model.train_batch`` (model/clip_tree.py:222-316) - as torch autograd over the functional towers of
``oracle/clip_ref.py``.

Only ``tests/`` may import this module.  It is pinned by the fixtures ``tools/make_golden.py`` captured from the
reference's own ``train_batch`` (tests/golden/train_*.npz + the ``train`` block of tree_*.json: loss, every
parameter's gradient norm, selected full gradients, the sampled negatives of every inner step):
``tests/test_oracle.py::test_train_ref_matches_reference_fixture``.  With that pin it serves as the checker of the
HIP training step at sizes the fixtures cannot hold (true-dimension ViT-L/14 + CoOp context, BASELINE configs[4]).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import clip_ref


def om_step(sd: Dict[str, torch.Tensor], images: torch.Tensor, node_tokens: torch.Tensor,
            picks: Sequence[Tuple[Sequence[int], int]], weights: Sequence[float], rd=clip_ref.identity,
            ctx: Optional[torch.Tensor] = None):
    """One OM / hierarchical step given the host-side decisions (negative lists ``picks`` = [(node ids, label
    position)] and the scalar weight of every inner step - clip_tree.py:256-259,265-275 produce them).

    Follows the reference's dataflow (clip_tree.py:224-226,261-280): image features once, L2-normalised, detached
    into a leaf; per inner step text-encode the picked prompts, normalise, ``logits = img_ @ text.T * exp(logit_scale)``,
    mean cross-entropy (all rows share the label) times the weight, backward; finally ONE image-tower backward from
    the leaf's accumulated gradient.  Returns (summed loss as float, {parameter name: gradient}, list of per-step CE)."""
    p = {k: (v.detach().clone().float().requires_grad_(True) if v.is_floating_point() and "running_" not in k and "num_batches" not in k
             else v) for k, v in sd.items()}
    c = ctx.detach().clone().float().requires_grad_(True) if ctx is not None else None
    img_feats = clip_ref.encode_image(p, images, rd)
    img_feats = img_feats / img_feats.norm(dim=-1, keepdim=True)                      # clip_tree.py:225
    img_leaf = img_feats.detach().clone().requires_grad_(True)                        # :226
    total, ces = 0.0, []
    for (ids, pos), w in zip(picks, weights):
        tok = node_tokens[torch.as_tensor(list(ids), dtype=torch.long)]
        tf = clip_ref.encode_text(p, tok, rd, trim=True, ctx=c)                       # :261 (trim: exact)
        tf = tf / tf.norm(dim=-1, keepdim=True)                                       # :262
        logits = (img_leaf @ tf.t()) * p["logit_scale"].exp()                         # :263
        labels = torch.full((images.shape[0],), int(pos), dtype=torch.long)
        ce = F.cross_entropy(logits, labels)                                          # :265
        loss_j = ce * float(w)
        loss_j.backward()                                                             # :276
        total += float(loss_j.detach())                                                        # :277
        ces.append(float(ce.detach()))
    img_feats.backward(img_leaf.grad)                                                 # :280
    grads = {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}
    if c is not None:
        grads["ctx"] = c.grad
    return total, grads, ces
