"""DGP baseline graph propagation on MI355X: the repository's ancestor / descendant aggregation,
`GCN_Dense_Att` of baseline/DGP/models/gcn_dense_att.py, with the same constructor, parameter names and `state_dict`
schema ('a_att', 'r_att', 'conv1.w', 'conv1.b', ..., 'conv-last.w', 'conv-last.b').

The reference keeps one torch sparse COO matrix per distance group and side and runs D `torch.mm(adj, support)` per
layer.  Here each side is ONE CSR over all groups (edge -> group id, 1 / degree); a layer is one fp32 MFMA product
(`hgr_matmul_f32`) and one gather-reduce launch (`hgr_csr_group_aggregate`) that also applies the attention weights,
the bias, LeakyReLU and the final row normalisation.  Inference only (`torch.no_grad`): the baseline's Adam training
loop (train_gcn_dense_att.py) is not part of this build.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from .. import ops

CHUNK = 256          # edges per work item: bounds the longest gather a single workgroup performs


def group_edges(n: int, edges: Sequence[Sequence[int]]) -> List[List[Tuple[int, int]]]:
    """materials/make_dense_grouped_graph.py:14-38: edges_set[d] = every (u, x) with x at BFS distance d from u along
    the directed edges (d = 0 holds the self pairs).  Same pair order as the reference script."""
    adjs: List[List[int]] = [[] for _ in range(n)]
    for u, v in edges:
        adjs[u].append(v)
    groups: List[List[Tuple[int, int]]] = []
    for u in range(n):
        dist = {u: 0}
        q, head = [u], 0
        while head < len(q):
            x = q[head]
            head += 1
            for y in adjs[x]:
                if y not in dist:
                    dist[y] = dist[x] + 1
                    q.append(y)
        for x, d in dist.items():
            while len(groups) <= d:
                groups.append([])
            groups[d].append((u, x))
    return groups


def fold_groups(edges_set: Sequence[Sequence[Sequence[int]]], lim: int = 4) -> List[List[Tuple[int, int]]]:
    """train_gcn_dense_att.py:52-56: distances beyond `lim` share group `lim`."""
    out = [[tuple(e) for e in g] for g in edges_set[:lim + 1]]
    for g in edges_set[lim + 1:]:
        out[lim].extend(tuple(e) for e in g)
    return out


class GraphOperator:
    """One side of the propagation: the D in-degree-normalised operators `normt_spm(adj_d or adj_d^T, 'in')`
    (baseline/DGP/utils.py:56-65) merged into a CSR with per-edge group ids, plus the work-item tables of the kernel."""

    def __init__(self, n: int, edges_set: Sequence[Sequence[Sequence[int]]], transpose: bool, device):
        assert len(edges_set) <= 32
        rows, cols, grps = [], [], []
        for d, edges in enumerate(edges_set):
            e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
            # a_adj = (adj^T row-normalised): row = edge target, column = edge source; r_adj: the other way round
            r, c = (e[:, 0], e[:, 1]) if transpose else (e[:, 1], e[:, 0])
            rows.append(r); cols.append(c); grps.append(np.full(len(e), d, np.int64))
        row, col, grp = np.concatenate(rows), np.concatenate(cols), np.concatenate(grps)
        # 1 / (row sum of the group's matrix) = 1 / number of edges of (row, group), duplicates counted like coo -> csr does
        key = row * len(edges_set) + grp
        cnt = np.bincount(key, minlength=n * len(edges_set))
        inv = (1.0 / cnt[key]).astype(np.float32)
        order = np.lexsort((col, grp, row))                     # by row, then group, then column: fixed summation order
        row, col, grp, inv = row[order], col[order], grp[order], inv[order]
        ptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(row, minlength=n), out=ptr[1:])
        item_row, item_e0, item_e1, item_slot = [], [], [], []
        split_row, split_slot0, split_n = [], [], []
        slots = 0
        for i in range(n):
            e0, e1 = int(ptr[i]), int(ptr[i + 1])
            k = max(1, -(-(e1 - e0) // CHUNK))
            if k == 1:
                item_row.append(i); item_e0.append(e0); item_e1.append(e1); item_slot.append(-1)
            else:
                split_row.append(i); split_slot0.append(slots); split_n.append(k)
                for j in range(k):
                    item_row.append(i); item_e0.append(e0 + j * CHUNK); item_e1.append(min(e1, e0 + (j + 1) * CHUNK)); item_slot.append(slots)
                    slots += 1
        # heavy items first: the tail of the launch is made of short rows
        order = np.argsort(-(np.asarray(item_e1) - np.asarray(item_e0)), kind="stable")
        i32 = lambda a: torch.tensor(np.asarray(a, dtype=np.int32), device=device)
        self.n, self.D, self.nnz, self.n_slots = n, len(edges_set), len(col), slots
        self.item_row, self.item_e0 = i32(np.asarray(item_row)[order]), i32(np.asarray(item_e0)[order])
        self.item_e1, self.item_slot = i32(np.asarray(item_e1)[order]), i32(np.asarray(item_slot)[order])
        self.split_row, self.split_slot0, self.split_n = i32(split_row), i32(split_slot0), i32(split_n)
        self.col = i32(col)
        self.inv_deg = torch.tensor(inv, device=device)
        self.grp = torch.tensor(grp.astype(np.uint8), device=device)
        self._partial = None

    def partial(self, c: int) -> torch.Tensor:
        if self.n_slots == 0:
            return None
        if self._partial is None or self._partial.shape[1] != c:
            self._partial = torch.empty((self.n_slots, c), dtype=torch.float32, device=self.col.device)
        return self._partial

    def aggregate(self, support: torch.Tensor, att: torch.Tensor, bias, out: torch.Tensor, slope: float, normalize: bool) -> torch.Tensor:
        ops.csr_group_aggregate(support, self, att, bias, out, slope, normalize)
        return out


class GraphConv(nn.Module):
    """gcn_dense_att.py:12-46: parameters `w [in, out]` (Xavier) and `b [out]`; LeakyReLU(0.2) unless `relu=False`."""

    def __init__(self, in_channels: int, out_channels: int, dropout: bool = False, relu: bool = True):
        super().__init__()
        self.dropout = nn.Dropout(p=0.5) if dropout else None
        self.w = nn.Parameter(torch.empty(in_channels, out_channels))
        self.b = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.w)
        self.relu = nn.LeakyReLU(negative_slope=0.2) if relu else None

    @torch.no_grad()
    def forward(self, inputs: torch.Tensor, op: GraphOperator, att: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        if self.training and self.dropout is not None:
            raise NotImplementedError("training-mode dropout: this build runs the DGP propagation for inference (model.eval())")
        n, c = inputs.shape[0], self.w.shape[1]
        if c % 4:
            raise ValueError(f"out_channels={c} must be a multiple of 4")
        support = torch.empty((n, c), dtype=torch.float32, device=inputs.device)
        ops.matmul_f32(inputs, self.w, support)                                    # bias is folded into the aggregation
        out = torch.empty((n, c), dtype=torch.float32, device=inputs.device)
        return op.aggregate(support, att, self.b, out, 0.2 if self.relu is not None else 1.0, normalize)


class GCN_Dense_Att(nn.Module):
    """gcn_dense_att.py:49-115.  `hidden_layers` like the reference: 'd2048,d' = dropout + 2048 hidden, dropout last."""

    def __init__(self, n: int, edges_set, in_channels: int, out_channels: int, hidden_layers: str, device="cuda"):
        super().__init__()
        self.n, self.d = n, len(edges_set)
        self.a_op = GraphOperator(n, edges_set, transpose=False, device=device)
        self.r_op = GraphOperator(n, edges_set, transpose=True, device=device)
        hl = hidden_layers.split(",")
        dropout_last = hl[-1] == "d"
        if dropout_last:
            hl = hl[:-1]
        self.a_att = nn.Parameter(torch.ones(self.d))
        self.r_att = nn.Parameter(torch.ones(self.d))
        layers, last_c, i = [], in_channels, 0
        for c in hl:
            dropout = c[0] == "d"
            c = int(c[1:] if dropout else c)
            i += 1
            conv = GraphConv(last_c, c, dropout=dropout)
            self.add_module("conv{}".format(i), conv)
            layers.append(conv)
            last_c = c
        conv = GraphConv(last_c, out_channels, relu=False, dropout=dropout_last)
        self.add_module("conv-last", conv)
        layers.append(conv)
        self.layers = layers
        self.to(device)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("GCN_Dense_Att needs device tensors: the product path has no CPU fallback")
        x = x.float().contiguous()
        graph_side = True
        for k, conv in enumerate(self.layers):
            op, att = (self.a_op, self.a_att) if graph_side else (self.r_op, self.r_att)
            att = torch.softmax(att.float(), dim=0).contiguous()                  # D <= 32 scalars
            x = conv(x, op, att, normalize=(k == len(self.layers) - 1))          # F.normalize fused into the last layer
            graph_side = not graph_side
        return x
