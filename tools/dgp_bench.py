#!/usr/bin/env python3
"""Dev tool: DGP propagation at ImageNet-21K scale (synthetic 12-level DAG, 300 -> 2048 -> 2048 channels): time the
merged-CSR kernel per layer, price it against the HBM roof, and time the reference's formulation (D torch.sparse.mm per
layer) on the same GPU and scipy on the host.  usage: dgp_bench.py [n_nodes]"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from hgr_net_amd import synth, ops
from hgr_net_amd.baseline import dgp, GCN_Dense_Att

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
dag = synth.make_dag(n, 12, seed=7, multi_parent=0.03)
names = ["fall11"] + sorted({w for e in dag for w in e} - {"fall11"})
idx = {w: i for i, w in enumerate(names)}
edges = [(idx[p], idx[c]) for p, c in dag]
n = len(names)
t0 = time.perf_counter(); groups = dgp.group_edges(n, edges); t_group = time.perf_counter() - t0
es = dgp.fold_groups(groups, 4)
t0 = time.perf_counter(); m = GCN_Dense_Att(n, es, 300, 2048, "d2048,d").eval(); t_build = time.perf_counter() - t0
x = torch.nn.functional.normalize(torch.randn(n, 300, device="cuda"))
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3
res = {"n": n, "pairs": [len(g) for g in es], "group_edges_s": round(t_group, 2), "build_s": round(t_build, 2)}
res["forward_ms"] = round(timeit(lambda: m(x)) * 1e3, 3)
sup = torch.randn(n, 2048, device="cuda"); out = torch.empty_like(sup)
for name, op, att in (("a_side", m.a_op, torch.softmax(m.a_att, 0)), ("r_side", m.r_op, torch.softmax(m.r_att, 0))):
    t = timeit(lambda: ops.csr_group_aggregate(sup, op, att, m.layers[0].b, out, 0.2, False))
    byts = (op.nnz + n) * 2048 * 4 + op.nnz * 9 + 2 * op.n_slots * 2048 * 4
    res[name] = {"us": round(t * 1e6, 1), "nnz": op.nnz, "items": op.item_row.numel(), "split_rows": op.split_row.numel(),
                 "gather_GBps": round(byts / t / 1e9), "frac_hbm_8TBps": round(byts / t / 8e12, 3)}
# the reference's formulation on the same GPU: D sparse COO matmuls per layer
def coo(edges, transpose):
    e = np.asarray(edges, np.int64)
    r, c = (e[:, 0], e[:, 1]) if transpose else (e[:, 1], e[:, 0])
    cnt = np.bincount(r, minlength=n)
    v = (1.0 / cnt[r]).astype(np.float32)
    return torch.sparse_coo_tensor(torch.from_numpy(np.vstack([r, c])), torch.from_numpy(v), (n, n)).coalesce().cuda()
a_set = [coo(e, False) for e in es]
att = torch.softmax(m.a_att, 0)
def ref_layer():
    o = None
    for i, adj in enumerate(a_set):
        y = torch.mm(adj, sup) * att[i]
        o = y if o is None else o + y
    return torch.nn.functional.leaky_relu(o, 0.2)
res["torch_sparse_a_side_us"] = round(timeit(ref_layer, 5) * 1e6, 1)
print(json.dumps(res))
