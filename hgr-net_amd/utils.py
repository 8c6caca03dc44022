"""Small host-side pieces of the reference's utils.py that the drivers use."""
from __future__ import annotations

import numpy as np

from .hierarchy import gen_tree  # noqa: F401  (utils.py:39-72)


def cosine_lr(optimizer, base_lrs, warmup_length, steps):
    """Linear warm-up then half-cosine decay, set on every param group (utils.py:82-95)."""
    if not isinstance(base_lrs, list):
        base_lrs = [base_lrs for _ in optimizer.param_groups]
    assert len(base_lrs) == len(optimizer.param_groups)

    def adjust(step):
        for group, base in zip(optimizer.param_groups, base_lrs):
            if step < warmup_length:
                lr = base * (step + 1) / warmup_length
            else:
                lr = 0.5 * (1 + np.cos(np.pi * (step - warmup_length) / (steps - warmup_length))) * base
            group["lr"] = lr

    return adjust


def count_acc(hits_dict, num_tot):
    """'Top@k(%):xx.xx, ...' string and the accuracy dict (utils.py:135-146)."""
    out, acc = "", {}
    keys = list(hits_dict.keys())
    for k in keys:
        acc[k] = hits_dict[k] / num_tot * 100.0
        out += "Top@{}(%):{:.2f}".format(k, acc[k])
        out += ", " if k != keys[-1] else "."
    return out, acc
