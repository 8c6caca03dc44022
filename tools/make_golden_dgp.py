#!/usr/bin/env python3
"""Generates tests/golden/dgp_small.npz by RUNNING the reference's DGP baseline code in this container:
baseline/DGP/materials/make_dense_grouped_graph.py (as a script, on a synthetic induced-graph JSON) and
baseline/DGP/models/gcn_dense_att.py::GCN_Dense_Att (eval mode, CPU: `.cuda()` patched to identity).  The fixture holds
inputs (graph, word vectors, parameters) and outputs (edge groups, per-layer activations, final vectors) only."""
import json
import os
import runpy
import sys
import tempfile
from pathlib import Path

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("HGR_REFERENCE", "/root/reference")) / "baseline" / "DGP"
sys.path.insert(0, str(REPO))

import numpy as np
import torch

from hgr_net_amd import synth
from oracle import dgp_ref

GOLD = REPO / "tests" / "golden"


def main():
    # a DAG with multi-parent nodes; induced-graph edges are (parent, child) index pairs over `wnids`
    n = 150
    dag = synth.make_dag(n, 7, seed=21, multi_parent=0.08)
    wnids = ["fall11"]
    for p, c in dag:
        for w in (p, c):
            if w not in wnids:
                wnids.append(w)
    idx = {w: i for i, w in enumerate(wnids)}
    edges = [[idx[p], idx[c]] for p, c in dag]
    n = len(wnids)
    dim_in, hidden, dim_out = 24, 40, 32
    vec = synth.normal(5, "dgp.vectors", n * dim_in).reshape(n, dim_in).astype(np.float32)

    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "induced.json"), os.path.join(tmp, "grouped.json")
        json.dump({"wnids": wnids, "vectors": vec.tolist(), "edges": edges}, open(src, "w"))
        argv = sys.argv
        sys.argv = ["make_dense_grouped_graph.py", "--input", src, "--output", dst]
        try:
            runpy.run_path(str(REF / "materials" / "make_dense_grouped_graph.py"), run_name="__main__")
        finally:
            sys.argv = argv
        grouped = json.load(open(dst))
    edges_set_full = grouped["edges_set"]
    mine = dgp_ref.group_edges(n, [tuple(e) for e in edges])
    assert [sorted(map(tuple, g)) for g in edges_set_full] == [sorted(g) for g in mine], "oracle grouping disagrees with the reference script"
    lim = 4
    edges_set = [list(g) for g in edges_set_full]
    for i in range(lim + 1, len(edges_set)):                       # train_gcn_dense_att.py:52-56
        edges_set[lim].extend(edges_set[i])
    edges_set = edges_set[:lim + 1]
    print("groups", [len(g) for g in edges_set_full], "->", [len(g) for g in edges_set])

    sys.path.insert(0, str(REF))
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    from models.gcn_dense_att import GCN_Dense_Att
    torch.manual_seed(3)
    gcn = GCN_Dense_Att(n, edges_set, dim_in, dim_out, f"d{hidden},d")
    with torch.no_grad():
        gcn.a_att.copy_(torch.tensor([0.3, -0.2, 0.5, 0.1, -0.4]))
        gcn.r_att.copy_(torch.tensor([-0.1, 0.4, 0.0, 0.25, 0.6]))
        for conv in gcn.layers:
            conv.b.copy_(torch.from_numpy(synth.normal(6, f"dgp.b{conv.b.numel()}", conv.b.numel()).astype(np.float32) * 0.1))
    gcn.eval()
    x = torch.nn.functional.normalize(torch.from_numpy(vec))        # train_gcn_dense_att.py:58-59
    taps = []
    with torch.no_grad():
        out = gcn(x)
        # per-layer activations, re-running the reference layers one at a time
        h, side = x, True
        for conv in gcn.layers:
            adj_set, att = (gcn.a_adj_set, gcn.a_att) if side else (gcn.r_adj_set, gcn.r_att)
            h = conv(h, adj_set, torch.softmax(att, 0))
            taps.append(h.numpy().copy())
            side = not side
    sd = {k: v.detach().numpy().copy() for k, v in gcn.state_dict().items()}
    layers = [(sd["conv1.w"], sd["conv1.b"], True), (sd["conv-last.w"], sd["conv-last.b"], False)]
    mine_out = dgp_ref.forward(x.numpy(), edges_set, layers, sd["a_att"], sd["r_att"])
    err = float(np.abs(mine_out - out.numpy()).max())
    print("oracle vs reference: max |diff| =", err)
    assert err < 2e-6
    np.savez_compressed(GOLD / "dgp_small.npz", wnids=np.array(wnids), edges=np.array(edges, np.int32), vectors=vec,
                        edges_set_sizes=np.array([len(g) for g in edges_set_full], np.int32),
                        edges_set_flat=np.array([e for g in edges_set_full for e in g], np.int32), lim=np.int32(lim),
                        x=x.numpy(), hidden=np.int32(hidden), tap0=taps[0], tap1=taps[1], out=out.numpy(),
                        **{"sd_" + k: v for k, v in sd.items()})
    print("state_dict keys:", list(sd), "wrote", GOLD / "dgp_small.npz", (GOLD / "dgp_small.npz").stat().st_size)


if __name__ == "__main__":
    main()
