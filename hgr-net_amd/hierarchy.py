"""Class-hierarchy construction.

Restates what the reference's ``gen_tree`` (utils.py:39-72) computes from the edge-list JSON, with
dict/array lookups instead of the reference's O(N^2) ``list.index`` scans:

* ``nodes``    - wnids in first-appearance order of the edge list, root ``fall11`` removed
                 (utils.py:43-45: ``nx.DiGraph.add_edges_from`` keeps insertion order);
* ``p2c[i]``   - children ids of node i in edge order (utils.py:48-51);
* ``c2p[i]``   - ids strictly between root and node i on the path
                 ``nx.shortest_path(G, 'fall11', node)`` (utils.py:53-56).  networkx answers that call
                 with a bidirectional BFS whose tie-breaking between equally short paths depends on
                 fringe sizes and adjacency insertion order; `_bidir_path` below follows the same
                 published procedure (networkx 3.4 ``bidirectional_shortest_path``) so that
                 multi-parent nodes get the same ancestor chain;
* ``d2n[d]``   - node ids whose chain has length d, in id order (utils.py:66-70);
* ``start_up`` - ids of the root's children (utils.py:46).

Also builds the flat arrays the device kernels consume: per-node depth, and a CSR of the levels.
"""
from __future__ import annotations

import json
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

ROOT = "fall11"


def _bidir_path(succ_adj: List[List[int]], pred_adj: List[List[int]], source: int, target: int) -> List[int]:
    if source == target:
        return [source]
    pred = {source: None}
    succ = {target: None}
    fwd, rev = [source], [target]
    meet = None
    while fwd and rev and meet is None:
        if len(fwd) <= len(rev):
            level, fwd = fwd, []
            for v in level:
                for w in succ_adj[v]:
                    if w not in pred:
                        fwd.append(w)
                        pred[w] = v
                    if w in succ:
                        meet = w
                        break
                if meet is not None:
                    break
        else:
            level, rev = rev, []
            for v in level:
                for w in pred_adj[v]:
                    if w not in succ:
                        succ[w] = v
                        rev.append(w)
                    if w in pred:
                        meet = w
                        break
                if meet is not None:
                    break
    if meet is None:
        raise ValueError(f"no path from root to node {target}")
    path = []
    w = meet
    while w is not None:
        path.append(w)
        w = pred[w]
    path.reverse()
    w = succ[path[-1]]
    while w is not None:
        path.append(w)
        w = succ[w]
    return path


@dataclass
class Hierarchy:
    nodes: List[str]
    p2c: List[List[int]]
    c2p: List[List[int]]
    d2n: Dict[int, List[int]]
    start_up: List[int]
    depth: np.ndarray = field(default=None)          # int32 [N]  = len(c2p[i])
    level_ptr: np.ndarray = field(default=None)      # int32 [max_depth + 2]  CSR over levels
    level_nodes: np.ndarray = field(default=None)    # int32 [N]   node ids grouped by level

    @property
    def max_depth(self) -> int:
        return max(self.d2n.keys())

    def as_tuple(self):
        """The reference's return order (utils.py:72)."""
        return self.p2c, self.c2p, self.d2n, self.nodes, self.start_up


def build_hierarchy(graph_edges: Sequence[Sequence[str]]) -> Hierarchy:
    index: Dict[str, int] = {}
    names: List[str] = []
    succ_adj: List[List[int]] = []
    pred_adj: List[List[int]] = []
    seen_edge = set()

    def nid(name: str) -> int:
        i = index.get(name)
        if i is None:
            i = len(names)
            index[name] = i
            names.append(name)
            succ_adj.append([])
            pred_adj.append([])
        return i

    for u, v in graph_edges:
        a, b = nid(u), nid(v)
        if (a, b) not in seen_edge:          # DiGraph keeps one edge per ordered pair
            seen_edge.add((a, b))
            succ_adj[a].append(b)
            pred_adj[b].append(a)
    if ROOT not in index:
        raise ValueError("edge list has no 'fall11' root")
    root = index[ROOT]
    # ids after removing the root from the node list (utils.py:45)
    remap = np.empty(len(names), dtype=np.int64)
    k = 0
    for i in range(len(names)):
        if i == root:
            remap[i] = -1
        else:
            remap[i] = k
            k += 1
    nodes = [n for i, n in enumerate(names) if i != root]
    start_up = [int(remap[c]) for c in succ_adj[root]]
    p2c = [[int(remap[c]) for c in succ_adj[i]] for i in range(len(names)) if i != root]
    c2p: List[List[int]] = []
    for i in range(len(names)):
        if i == root:
            continue
        path = _bidir_path(succ_adj, pred_adj, root, i)
        c2p.append([int(remap[p]) for p in path[1:-1]])
    # the reference asserts consecutive chain members are parent/child (utils.py:58-64)
    for chain in c2p:
        for a, b in zip(chain[:-1], chain[1:]):
            assert b in p2c[a]
    d2n: Dict[int, List[int]] = defaultdict(list)
    for i, chain in enumerate(c2p):
        d2n[len(chain)].append(i)
    depth = np.array([len(c) for c in c2p], dtype=np.int32)
    maxd = int(depth.max()) if len(depth) else 0
    level_ptr = np.zeros(maxd + 2, dtype=np.int32)
    for d in range(maxd + 1):
        level_ptr[d + 1] = level_ptr[d] + len(d2n.get(d, []))
    level_nodes = np.concatenate([np.array(d2n.get(d, []), dtype=np.int32) for d in range(maxd + 1)]) if len(depth) else np.zeros(0, np.int32)
    return Hierarchy(nodes, p2c, c2p, d2n, start_up, depth, level_ptr, level_nodes)


def gen_tree(opts) -> tuple:
    """Same signature and return value as the reference's ``utils.gen_tree`` (utils.py:39-72)."""
    with open(opts.graph_path, "r") as f:
        return build_hierarchy(json.load(f)).as_tuple()
