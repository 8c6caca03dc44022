"""DGP baseline graph propagation: oracle and host logic against the fixture produced by the
reference's own script + module (tools/make_golden_dgp.py); the HIP path against both."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import dgp_ref

GOLD = Path(__file__).parent / "golden" / "dgp_small.npz"


def _fixture():
    z = np.load(GOLD)
    sizes, flat = z["edges_set_sizes"], z["edges_set_flat"]
    groups, p = [], 0
    for s in sizes:
        groups.append([tuple(e) for e in flat[p:p + s].tolist()])
        p += s
    return z, groups


def test_oracle_grouping_and_forward_match_reference():
    z, groups = _fixture()
    n = len(z["wnids"])
    mine = dgp_ref.group_edges(n, [tuple(e) for e in z["edges"].tolist()])
    assert mine == groups                                            # same pairs, same order as the reference script
    es = dgp_ref.fold_groups(groups, int(z["lim"]))
    layers = [(z["sd_conv1.w"], z["sd_conv1.b"], True), (z["sd_conv-last.w"], z["sd_conv-last.b"], False)]
    out = dgp_ref.forward(z["x"], es, layers, z["sd_a_att"], z["sd_r_att"])
    assert np.abs(out - z["out"]).max() < 2e-6
    assert np.allclose(np.linalg.norm(out, axis=1)[np.abs(out).sum(1) > 0], 1.0, atol=1e-5)


def test_host_grouping_matches_reference():
    from hgr_net_amd.baseline import dgp
    z, groups = _fixture()
    mine = dgp.group_edges(len(z["wnids"]), z["edges"].tolist())
    assert mine == groups
    folded = dgp.fold_groups(groups, int(z["lim"]))
    assert len(folded) == int(z["lim"]) + 1 and sum(map(len, folded)) == sum(map(len, groups))
    assert folded == dgp_ref.fold_groups(groups, int(z["lim"]))


def _model(z, es, device):
    from hgr_net_amd.baseline import GCN_Dense_Att
    m = GCN_Dense_Att(len(z["wnids"]), es, z["x"].shape[1], z["out"].shape[1], f"d{int(z['hidden'])},d", device=device)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_")}
    assert set(m.state_dict()) == set(sd)                            # the reference's state_dict schema
    m.load_state_dict(sd)
    return m.eval()


@pytest.mark.gpu
def test_gcn_dense_att_matches_reference_fixture():
    z, groups = _fixture()
    es = dgp_ref.fold_groups(groups, int(z["lim"]))
    m = _model(z, es, "cuda")
    x = torch.from_numpy(z["x"]).cuda()
    out = m(x).cpu().numpy()
    assert np.abs(out - z["out"]).max() < 2e-6, np.abs(out - z["out"]).max()
    # layer by layer: the fixture's activations after conv1 (LeakyReLU) and conv-last (before F.normalize)
    att = torch.softmax(m.a_att, 0)
    h = m.layers[0](x, m.a_op, att)
    assert np.abs(h.cpu().numpy() - z["tap0"]).max() < 5e-6
    h2 = m.layers[1](h, m.r_op, torch.softmax(m.r_att, 0))
    assert np.abs(h2.cpu().numpy() - z["tap1"]).max() < 5e-6
    with pytest.raises(NotImplementedError):
        m.train()(x)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [64, 1028, 2048])
def test_skewed_graph_split_rows_match_oracle(c):
    """A deep, wide DAG: the root's descendant list is far longer than one work item (rows cut into partials), many rows
    have empty groups, channel counts that need 1 and 2 column passes and a ragged tail."""
    from hgr_net_amd import synth
    from hgr_net_amd.baseline import GraphOperator, dgp
    from hgr_net_amd import ops
    n = 1500
    dag = synth.make_dag(n, 9, seed=4, multi_parent=0.05)
    names = ["fall11"] + sorted({w for e in dag for w in e} - {"fall11"})
    idx = {w: i for i, w in enumerate(names)}
    edges = [(idx[p], idx[ch]) for p, ch in dag]
    n = len(names)
    es = dgp.fold_groups(dgp.group_edges(n, edges), 4)
    rng = np.random.default_rng(c)
    support = rng.standard_normal((n, c)).astype(np.float32)
    bias = rng.standard_normal(c).astype(np.float32)
    att = dgp_ref.softmax(rng.standard_normal(len(es)).astype(np.float32))
    for transpose in (False, True):
        op = GraphOperator(n, es, transpose=transpose, device="cuda")
        if transpose:
            assert op.n_slots > 0 and op.split_row.numel() > 0       # the root row is split
        adj = [dgp_ref.norm_in(n, e, transpose=transpose).astype(np.float64) for e in es]
        ref = sum((a @ (support.astype(np.float64) + bias)) * w for a, w in zip(adj, att.astype(np.float64)))
        for slope, normalize in ((0.2, False), (1.0, True)):
            want = np.where(ref >= 0, ref, ref * slope)
            if normalize:
                want = want / np.maximum(np.sqrt((want * want).sum(1, keepdims=True)), 1e-12)
            out = torch.empty((n, c), dtype=torch.float32, device="cuda")
            ops.csr_group_aggregate(torch.from_numpy(support).cuda(), op, torch.from_numpy(att).cuda(), torch.from_numpy(bias).cuda(),
                                    out, slope, normalize)
            got = out.cpu().numpy()
            assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max()), (transpose, slope, normalize)
            out2 = torch.empty_like(out)
            ops.csr_group_aggregate(torch.from_numpy(support).cuda(), op, torch.from_numpy(att).cuda(), torch.from_numpy(bias).cuda(),
                                    out2, slope, normalize)
            assert torch.equal(out, out2)                            # deterministic: no atomics
