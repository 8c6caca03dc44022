"""ORACLE (test infrastructure, never a product path): CPU restatement of the hierarchy-aware
zero-shot scoring of the reference - ``tree_model.update_classifier`` / ``forward``
(model/clip_tree.py:318-333) and the metric loop of ``main.test`` (main.py:131-203).

Plain numpy / Python loops, written for small inputs.  Checked against the real ``main.test`` run in
the build container (tools/make_golden.py -> tests/golden/eval_*.npz).  Index results are compared
bit-exactly; ties are broken towards the lowest column index (torch.topk leaves tie order
unspecified, fixtures avoid exact ties).
"""
from __future__ import annotations

import copy
from typing import Dict, List, Sequence

import numpy as np
import torch

from . import clip_ref

TOPK = (1, 2, 5, 10, 20)  # main.py:121


def update_classifier(sd, node_tokens: torch.Tensor, rd=clip_ref.identity, trim: bool = False) -> torch.Tensor:
    """Text-encode all node prompts in two halves, concatenate, L2-normalise rows
    (model/clip_tree.py:318-325)."""
    n = node_tokens.shape[0]
    with torch.no_grad():
        t1 = clip_ref.encode_text(sd, node_tokens[: n // 2], rd, trim)
        t2 = clip_ref.encode_text(sd, node_tokens[n // 2:], rd, trim)
        t = torch.cat([t1, t2])
        return t / t.norm(dim=-1, keepdim=True)


def forward(sd, images: torch.Tensor, zsl_weights: torch.Tensor, rd=clip_ref.identity) -> torch.Tensor:
    """encode_image -> row L2-normalise -> feats @ zsl_weights.T, no temperature
    (model/clip_tree.py:328-333)."""
    with torch.no_grad():
        f = clip_ref.encode_image(sd, images, rd)
        f = f / f.norm(dim=-1, keepdim=True)
        return rd(f) @ rd(zsl_weights.float()).t()


def topk_desc(row: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries, largest first, ties to the lowest index."""
    return np.argsort(-row, kind="stable")[:k]


def level_argmax(logits: np.ndarray, train_index: np.ndarray, level_nodes: Sequence[int], n_nodes: int) -> np.ndarray:
    """main.py:164-176 for one ancestor level: fill every column outside the level with -1, restrict to
    ``train_index`` columns, arg-max, map back to node ids."""
    masked = logits.copy()
    rest = np.array(sorted(set(range(n_nodes)) - set(level_nodes)), dtype=np.int64)
    if rest.size:
        masked[:, rest] = -1.0
    sub = masked[:, train_index]
    return train_index[np.array([topk_desc(r, 1)[0] for r in sub])]


class EvalState:
    """The counters ``main.test`` keeps across batches (main.py:121-128)."""

    def __init__(self):
        self.hits = {k: 0.0 for k in TOPK}
        self.num_sample = 0
        self.hits_all = 0.0
        self.path_all = 0.0
        self.point_all = 0.0

    def add_batch(self, logits: np.ndarray, target: int, c2p, d2n, train_index: np.ndarray, test_index: np.ndarray):
        """One iteration of the loop at main.py:131-191.  Returns (pred_top20 [B,20], dict_path [B,L])."""
        b, n = logits.shape
        # T1 - top-k hits over the test columns (main.py:136-148)
        sub = logits[:, test_index]
        pred = test_index[np.stack([topk_desc(r, max(TOPK)) for r in sub])]
        correct = pred == target
        for k in TOPK:
            self.hits[k] += float(correct[:, :k].sum())
        self.num_sample += b
        # T2 - top-1 over the train columns vs {ancestors, target} (main.py:152-160)
        parents = copy.copy(c2p[target]) + [target]
        sub = logits[:, train_index]
        p1 = train_index[np.array([topk_desc(r, 1)[0] for r in sub])]
        self.hits_all += float(sum(int(p == q) for p in p1 for q in parents))
        # T3 - per-ancestor-level masked arg-max (main.py:162-176)
        dict_path = np.zeros((b, len(parents)), dtype=np.float32)
        for k, p in enumerate(parents):
            level = len(c2p[p])
            same_l = copy.copy(d2n[level])
            if p not in same_l:
                same_l.append(p)
            dict_path[:, k] = level_argmax(logits, train_index, same_l, n)
        # T4 - point / edge overlap (main.py:177-191)
        edge = 0
        point = 0
        L = len(parents)
        for i in range(b):
            if L - 1 == 0 and parents[0] == dict_path[i][0]:
                self.path_all += 1
            for j in range(L - 1):
                if parents[j] == dict_path[i][j]:
                    point += 1
                if parents[j] == dict_path[i][j] and parents[j + 1] == dict_path[i][j + 1]:
                    edge += 1
            if parents[L - 1] == dict_path[i][L - 1]:
                point += 1
        if L - 1 != 0:
            self.path_all += edge / (L - 1)
        self.point_all += point / L
        return pred, dict_path

    def summary(self) -> str:
        """The string ``main.test`` prints and logs (main.py:205-216, utils.py:135-146)."""
        out = "\n"
        keys = list(self.hits.keys())
        for k in keys:
            out += "Top@{}(%):{:.2f}".format(k, self.hits[k] / self.num_sample * 100.0)
            out += ", " if k != keys[-1] else "."
        out += " hit_ratio(%):{:.2f}".format(self.hits_all / self.num_sample * 100.0)
        out += " path_ratio(%):{:.2f}".format(self.path_all / self.num_sample * 100.0)
        out += " point_ratio(%):{:.2f}".format(self.point_all / self.num_sample * 100.0)
        return out

    def counters(self) -> Dict[str, float]:
        d = {f"hits@{k}": v for k, v in self.hits.items()}
        d.update(hits_all=self.hits_all, path_all=self.path_all, point_all=self.point_all, num_sample=float(self.num_sample))
        return d
