"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's DGP baseline graph propagation.
Never imported by the product path.

Follows, line by line:
  * baseline/DGP/materials/make_dense_grouped_graph.py:14-38   per-node BFS -> (u, x) pairs grouped by distance
  * baseline/DGP/train_gcn_dense_att.py:52-56                  groups beyond `lim` folded into group `lim`
  * baseline/DGP/utils.py:56-65 `normt_spm(method='in')`       transpose, divide every row by its sum (empty rows stay 0)
  * baseline/DGP/models/gcn_dense_att.py:31-46  `GraphConv.forward`   support = x W + b;  sum_d att_d * (A_d support); LeakyReLU(0.2)
  * baseline/DGP/models/gcn_dense_att.py:103-115 `GCN_Dense_Att.forward`  layers alternate the ancestor-side (a_adj, a_att)
    and descendant-side (r_adj, r_att) operators, softmax over the attention logits, row L2-normalise at the end
Pinned by tests/golden/dgp_*.npz, which tools/make_golden_dgp.py produced by running the reference's own script and
module on a synthetic graph.  Plain numpy, fp64 accumulation available for tolerance studies (`dtype`).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def group_edges(n: int, edges: Sequence[Tuple[int, int]]) -> List[List[Tuple[int, int]]]:
    """make_dense_grouped_graph.py:14-38: new_edges[dis] = all (u, x) with x at BFS distance dis from u (dis 0 = (u, u))."""
    adjs: Dict[int, List[int]] = {i: [] for i in range(n)}
    for u, v in edges:
        adjs[u].append(v)
    new_edges: List[List[Tuple[int, int]]] = [[] for _ in range(99)]
    for u in range(n):
        q, l, d = [u], 0, {u: 0}
        while l < len(q):
            x = q[l]
            l += 1
            for y in adjs[x]:
                if d.get(y) is None:
                    d[y] = d[x] + 1
                    q.append(y)
        for x, dis in d.items():
            new_edges[dis].append((u, x))
    while new_edges[-1] == []:
        new_edges.pop()
    return new_edges


def fold_groups(edges_set: List[List[Tuple[int, int]]], lim: int = 4) -> List[List[Tuple[int, int]]]:
    """train_gcn_dense_att.py:52-56."""
    edges_set = [list(e) for e in edges_set]
    for i in range(lim + 1, len(edges_set)):
        edges_set[lim].extend(edges_set[i])
    return edges_set[:lim + 1]


def norm_in(n: int, edges: Sequence[Tuple[int, int]], transpose: bool = False) -> np.ndarray:
    """Dense form of `normt_spm(adj, 'in')` for adj[u, v] = multiplicity of edge (u, v) (or of its transpose)."""
    adj = np.zeros((n, n), np.float64)
    for u, v in edges:
        if transpose:
            adj[v, u] += 1.0
        else:
            adj[u, v] += 1.0
    mx = adj.T
    rowsum = mx.sum(1)
    with np.errstate(divide="ignore"):
        r_inv = np.where(rowsum != 0, 1.0 / rowsum, 0.0)
    return (mx * r_inv[:, None]).astype(np.float32)


def softmax(v: np.ndarray) -> np.ndarray:
    e = np.exp(v - v.max())
    return e / e.sum()


def forward(x: np.ndarray, edges_set, layers, a_att: np.ndarray, r_att: np.ndarray, dtype=np.float32) -> np.ndarray:
    """`GCN_Dense_Att.forward` in eval mode (dropout off).  layers = [(w [in, out], b [out], relu: bool), ...]."""
    n = x.shape[0]
    a_adj = [norm_in(n, e).astype(dtype) for e in edges_set]
    r_adj = [norm_in(n, e, transpose=True).astype(dtype) for e in edges_set]
    x = x.astype(dtype)
    graph_side = True
    for w, b, relu in layers:
        adj_set, att = (a_adj, softmax(a_att.astype(dtype))) if graph_side else (r_adj, softmax(r_att.astype(dtype)))
        support = x @ w.astype(dtype) + b.astype(dtype)
        out = None
        for i, adj in enumerate(adj_set):
            y = (adj @ support) * att[i]
            out = y if out is None else out + y
        if relu:
            out = np.where(out >= 0, out, out * dtype(0.2))
        x = out
        graph_side = not graph_side
    norm = np.maximum(np.sqrt((x * x).sum(1, keepdims=True)), 1e-12)           # F.normalize: x / max(|x|, eps)
    return (x / norm).astype(np.float32)
