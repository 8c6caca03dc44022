"""CPU: the oracle (oracle/) against the fixtures captured from the reference's own code
(tools/make_golden.py), plus the synthetic generators' determinism."""
import json

import numpy as np
import pytest
import torch

from hgr_net_amd import synth
from hgr_net_amd.hierarchy import build_hierarchy
from oracle import clip_ref, tree_ref

CLIP_CASES = ["tiny-vit", "small-vit", "tiny-rn", "small-rn", "small-rnx", "ViT-B_32", "RN50", "RN50x4", "ViT-L_14"]   # RN101 / RN50x16 / ViT-B_16 fixtures: GPU suite only (same oracle code, slow on CPU)
TREE_CASES = ["tinyvit_n90", "smallvit_n300", "tinyrn_n64"]


def test_synth_is_bit_reproducible():
    a = synth.normal(3, "x", 1000)
    b = synth.normal(3, "x", 1000)
    assert np.array_equal(a, b)
    # pinned values: the generator must not drift between containers
    u = synth.uniform(0, "pin", 4)
    assert u.min() >= 0 and u.max() < 1
    sd1 = synth.clip_state_dict("tiny-vit", 0)
    sd2 = synth.clip_state_dict("tiny-vit", 0)
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    assert abs(float(synth.normal(0, "stat", 200000).std()) - 1.0) < 0.01


def test_infer_config_roundtrip():
    for name in ("tiny-vit", "tiny-rn", "small-vit"):
        cfg = synth.CLIP_CONFIGS[name]
        got = clip_ref.infer_config(synth.clip_state_dict(name, 0))
        for k, v in cfg.items():
            assert got[k] == v, (name, k, got[k], v)


@pytest.mark.parametrize("case", CLIP_CASES)
def test_clip_towers_match_reference_fixture(case, golden_dir):
    z = np.load(golden_dir / f"clip_{case}.npz")
    cfg = json.loads(str(z["config"]))
    if isinstance(cfg["vision_layers"], list):
        cfg["vision_layers"] = tuple(cfg["vision_layers"])
    sd = synth.clip_state_dict(cfg, int(z["seed"]))
    img = synth.images(int(z["batch"]), cfg["image_resolution"], int(z["image_seed"]))
    tok = synth.make_tokens(int(z["n_text"]), int(z["token_seed"]), cfg["vocab_size"])
    with torch.no_grad():
        fi = clip_ref.encode_image(sd, img).numpy()
        ft = clip_ref.encode_text(sd, tok).numpy()
        ft_trim = clip_ref.encode_text(sd, tok, trim=True).numpy()
    # fp32 reduction order differs between torch builds; 1e-5 relative to feature scale
    assert np.abs(fi - z["image_features"]).max() < 1e-5 * max(1.0, np.abs(z["image_features"]).max())
    assert np.abs(ft - z["text_features"]).max() < 1e-5
    assert np.abs(ft_trim - z["text_features"]).max() < 1e-5      # EOT trimming is exact (causal mask)


@pytest.mark.parametrize("case", TREE_CASES)
def test_hierarchy_matches_reference_gen_tree(case, golden_dir):
    meta = json.load(open(golden_dir / f"tree_{case}.json"))
    d = meta["dag"]
    h = build_hierarchy(synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"]))
    assert h.nodes == meta["nodes"]
    assert h.p2c == meta["p2c"] and h.c2p == meta["c2p"] and h.start_up == meta["start_up"]
    assert {str(k): v for k, v in h.d2n.items()} == meta["d2n"]
    assert [str(k) for k in h.d2n.keys()] == list(meta["d2n"].keys())     # first-appearance key order
    assert (h.depth == np.array([len(c) for c in h.c2p])).all()


def test_hierarchy_matches_networkx_tie_breaking():
    nx = pytest.importorskip("networkx")
    for seed in range(4):
        edges = synth.make_dag(400, depth=9, seed=seed, multi_parent=0.25)
        h = build_hierarchy(edges)
        g = nx.DiGraph()
        g.add_edges_from(edges)
        names = [n for n in g.nodes() if n != "fall11"]
        idx = {n: i for i, n in enumerate(names)}
        assert names == h.nodes
        for i, n in enumerate(names):
            chain = [idx[p] for p in nx.shortest_path(g, "fall11", n)[1:-1]]
            assert chain == h.c2p[i], (seed, n)


@pytest.mark.parametrize("case", TREE_CASES)
def test_tree_forward_and_metrics_match_reference(case, golden_dir):
    meta = json.load(open(golden_dir / f"tree_{case}.json"))
    z = np.load(golden_dir / f"tree_{case}.npz")
    cfg = meta["config"]
    if isinstance(cfg["vision_layers"], list):
        cfg["vision_layers"] = tuple(cfg["vision_layers"])
    sd = synth.clip_state_dict(cfg, 0)
    d = meta["dag"]
    h = build_hierarchy(synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"]))
    leaf = [len(c) == 0 for c in h.p2c]
    splits = synth.make_splits(h.nodes, leaf, meta["n_train"], meta["n_test"], meta["split_seed"])
    index = {n: i for i, n in enumerate(h.nodes)}
    train_index = np.array([index[n] for n in splits["all"]], dtype=np.int64)
    test_index = np.array([index[n] for n in splits["rest"]], dtype=np.int64)
    tokens = torch.from_numpy(z["node_tokens"].astype(np.int64))
    zsl = tree_ref.update_classifier(sd, tokens, trim=True)
    assert np.abs(zsl.numpy() - z["zsl_weights"]).max() < 2e-6
    st = tree_ref.EvalState()
    for i in range(meta["batches"]):
        img = synth.images(meta["bsz"], cfg["image_resolution"], meta["image_seed0"] + i)
        lg = tree_ref.forward(sd, img, zsl).numpy()
        assert np.abs(lg - z["logits"][i]).max() < 1e-5
        # metrics from the REFERENCE's logits so index results are compared bit-exactly
        pred, path = st.add_batch(z["logits"][i], meta["targets"][i], h.c2p, h.d2n, train_index, test_index)
        assert np.array_equal(pred, z["pred_top20"][i])
        assert np.array_equal(path, z[f"dict_path_{i}"])
    assert st.summary() == meta["metric"]                 # the exact string main.test printed
    for k, v in meta["counters"].items():
        assert abs(st.counters()[k] - v) < 1e-9


def test_bf16_emulation_error_budget():
    """bf16 MFMA inputs + fp32 residual keep the image-side logit error well inside 1e-3 on a small model."""
    sd = synth.clip_state_dict("small-vit", 0)
    tok = synth.make_tokens(64, 11, 1024)
    img = synth.images(4, 96, 5)
    zsl = tree_ref.update_classifier(sd, tok, trim=True)
    ref = tree_ref.forward(sd, img, zsl)
    emu = tree_ref.forward(sd, img, zsl, rd=clip_ref.round_bf16)
    assert float((ref - emu).abs().max()) < 1e-3


def test_coop_prompt_path_matches_reference_fixture(golden_dir):
    """oracle encode_text(ctx=...) vs the reference's PromptLearner + TextEncoder (model/CoOp.py) output."""
    z = np.load(golden_dir / "coop_tinyvit.npz")
    cfg = json.loads(str(z["config"]))
    sd = synth.clip_state_dict(cfg, 0)
    tok = torch.from_numpy(z["tokens"].astype(np.int64))[torch.from_numpy(z["idx"])]
    f = clip_ref.encode_text(sd, tok, trim=True, ctx=torch.from_numpy(z["ctx"])).numpy()
    assert np.abs(f - z["features"]).max() < 1e-5


# ---- image transform (oracle/resample_ref.py) -----------------------------------------------------
def test_resample_oracle_matches_golden(golden_dir):
    """The Pillow restatement against vectors produced by Pillow + torch (tools/make_golden_preproc.py)."""
    from oracle import resample_ref
    g = np.load(golden_dir / "preproc.npz")
    for i in range(int(g["n_cases"])):
        n = int(g[f"npx_{i}"])
        u8 = resample_ref.transform_u8(g[f"in_{i}"], n)
        assert np.array_equal(u8, g[f"u8_{i}"]), i
        assert np.array_equal(resample_ref.normalize(u8), g[f"f32_{i}"]), i


def test_resample_oracle_matches_pillow_live():
    """Same check against the Pillow installed next to the tests, on sizes the fixture does not hold."""
    Image = pytest.importorskip("PIL.Image")
    from oracle import resample_ref
    rng = np.random.default_rng(11)
    for h, w, n in [(375, 500, 224), (500, 333, 224), (97, 61, 48), (48, 48, 64), (31, 400, 32)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        nw, nh = resample_ref.resized_size(w, h, n)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BICUBIC))
        assert np.array_equal(resample_ref.resize_bicubic(img, nw, nh), ref), (h, w, n)


@pytest.mark.parametrize("case", ["tinyvit_n90", "tinyrn_n64"])
def test_train_ref_matches_reference_fixture(case, golden_dir, tmp_path):
    """oracle/train_ref.om_step (fp32 autograd over the functional towers) against what the reference's own
    train_batch produced on the same weights / images / sampled negatives: loss, EVERY parameter's gradient norm and
    the stored full gradients.  This pins the oracle used for the true-dimension ViT-L/14 + CoOp step on the GPU."""
    import types
    from hgr_net_amd.clip.model import build_model
    from hgr_net_amd.model import tree_model
    from oracle import train_ref
    meta = json.load(open(golden_dir / f"tree_{case}.json"))
    z = np.load(golden_dir / f"tree_{case}.npz")
    gold = np.load(golden_dir / f"train_{case}.npz")
    cfg, d, t = meta["config"], meta["dag"], meta["train"]
    if isinstance(cfg["vision_layers"], list):
        cfg["vision_layers"] = tuple(cfg["vision_layers"])
    edges = synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"])
    (tmp_path / "g.json").write_text(json.dumps(edges))
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    sd = synth.clip_state_dict(cfg, 0)
    o = types.SimpleNamespace(device="cpu", folder=str(tmp_path), exp_name="HGR", weights="equal", from_epoch=-1, graph_path=str(tmp_path / "g.json"),
                              arch="x", fetch=False, load=False, load_path="none", scale=1.0, **t["opts"])
    tokens = torch.from_numpy(z["node_tokens"].astype(np.int64))
    m = tree_model(o, splits["all"], splits["rest"], node_tokens=tokens, clip_model=build_model(sd))      # host logic only
    plan = m.outer_inner_plan(t["target"])
    weights = [float(m.get_weights("equal", st["M"])[st["m_loop"]] * m.get_weights("equal", st["K"])[st["k_loop"]]) for st in plan]
    img = synth.images(t["bsz"], cfg["image_resolution"], t["image_seed"])
    loss, grads, _ = train_ref.om_step(sd, img, tokens, [tuple(c) for c in t["contra"]], weights)
    assert abs(loss - t["loss"]) < 1e-4 * abs(t["loss"]), (loss, t["loss"])
    for k, ref in t["grad_norms"].items():
        got = float(grads[k].norm()) if k in grads else 0.0
        assert abs(got - ref) <= 2e-3 * ref + 1e-6, (k, got, ref)
    n_full = 0
    for key in gold.files:
        if key.startswith("grad/"):
            gref = torch.from_numpy(gold[key])
            assert float((grads[key[5:]] - gref).abs().max()) <= 2e-3 * float(gref.abs().max()) + 1e-7, key
            n_full += 1
    assert n_full >= 10
