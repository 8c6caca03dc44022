"""GPU: CLIP towers, tree_model forward and the evaluation metrics through the C ABI, against the
fixtures captured from the reference (tests/golden) and the CPU oracle."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hgr_net_amd import evaluate, ops, synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.model import tree_model
from oracle import clip_ref, tree_ref

DEV = "cuda"
# |feature| is O(1..3); bf16 MFMA inputs carry 2^-9 relative rounding per operand, f16 2^-12
FEAT_TOL = {"bf16": 4e-2, "f16": 6e-3}


def _cfg(z):
    cfg = json.loads(str(z["config"])) if not isinstance(z, dict) else z
    if isinstance(cfg["vision_layers"], list):
        cfg["vision_layers"] = tuple(cfg["vision_layers"])
    return cfg


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", ["tiny-vit", "small-vit", "ViT-B_32", "ViT-L_14", "tiny-rn", "small-rn", "RN50",
                                  # the rest of the reference's clip._MODELS (clip/clip.py:25-32) and a small width-48 tower:
                                  # RN widths that are not multiples of 64 are stored zero-padded (clip/model.py `_cpad`)
                                  "ViT-B_16", "small-rnx", "RN101", "RN50x4", "RN50x16"])
def test_towers_vs_reference_fixture(case, dt, golden_dir):
    z = np.load(golden_dir / f"clip_{case}.npz")
    cfg = _cfg(z)
    sd = synth.clip_state_dict(cfg, int(z["seed"]))
    model = build_model(sd, image_dtype=dt, text_dtype=dt).to(DEV)
    img = synth.images(int(z["batch"]), cfg["image_resolution"], int(z["image_seed"]))
    tok = synth.make_tokens(int(z["n_text"]), int(z["token_seed"]), cfg["vocab_size"])
    fi = model.encode_image(img.to(DEV)).cpu().numpy()
    ft = model.encode_text(tok.to(DEV)).cpu().numpy()
    ft_full = model.encode_text(tok.to(DEV), trim=False).cpu().numpy()
    assert np.abs(fi - z["image_features"]).max() < FEAT_TOL[dt] * max(1.0, np.abs(z["image_features"]).max())
    assert np.abs(ft - z["text_features"]).max() < FEAT_TOL[dt]
    assert np.abs(ft_full - z["text_features"]).max() < FEAT_TOL[dt]
    # cosine of HIP vs reference features ~ 1
    cos = (fi * z["image_features"]).sum(-1) / np.linalg.norm(fi, axis=-1) / np.linalg.norm(z["image_features"], axis=-1)
    # bf16 activations through RN50x16's 40 bottlenecks (8-bit mantissa per stored activation) measure 0.99944
    assert cos.min() > ((0.999 if case == "RN50x16" else 0.9995) if dt == "bf16" else 0.99999)


def test_vit_taps_match_oracle_layer_by_layer():
    """Per-layer residual stream vs the oracle run with the same 16-bit rounding points."""
    sd = synth.clip_state_dict("small-vit", 0)
    model = build_model(sd, image_dtype="f16").to(DEV)
    img = synth.images(2, 96, 77)
    taps, otaps = {}, {}
    model.encode_image(img.to(DEV), taps=taps)
    clip_ref.vit_forward(sd, img, clip_ref.round_f16, otaps)
    for k, v in otaps.items():
        assert (taps[k].cpu() - v).abs().max() < 5e-3 * max(1.0, float(v.abs().max())), k


def _tree_case(case, golden_dir, dt="bf16", tdt="f16"):
    meta = json.load(open(golden_dir / f"tree_{case}.json"))
    z = np.load(golden_dir / f"tree_{case}.npz")
    cfg = _cfg(meta["config"])
    d = meta["dag"]
    edges = synth.make_dag(meta["n_nodes"], d["depth"], d["seed"], d["multi_parent"])
    return meta, z, cfg, edges


def _opts(tmp_path, edges, **kw):
    g = tmp_path / "graph.json"
    g.write_text(json.dumps(edges))
    o = types.SimpleNamespace(device=DEV, folder=str(tmp_path / "out"), exp_name="HGR", weights="equal", out_ratio=0.25,
                              in_ratio=0.5, from_epoch=-1, graph_path=str(g), arch="synthetic", fetch=False, load=False,
                              load_path="none", scale=1.0, num_compare=256, k=1, sample_strategy="topk", weighting="both")
    o.__dict__.update(kw)
    return o


# north_star tolerance: fp32 logits within 1e-3 of the reference PyTorch path.  The default f16 towers meet
# it with margin; bf16 MFMA inputs (8 mantissa bits) land at ~1e-3 on these small, large-logit models and
# are held to their own measured bound - which is why bf16 is not the default (DESIGN.md "Precision").
LOGIT_TOL = {"f16": 1e-3, "bf16": 2.5e-3}


@pytest.mark.parametrize("idt", ["f16", "bf16"])
@pytest.mark.parametrize("case", ["tinyvit_n90", "smallvit_n300", "tinyrn_n64"])
def test_tree_model_forward_and_metrics_vs_reference(case, idt, golden_dir, tmp_path):
    meta, z, cfg, edges = _tree_case(case, golden_dir)
    sd = synth.clip_state_dict(cfg, 0)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    clip_model = build_model(sd, image_dtype=idt, text_dtype="f16").to(DEV)
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"],
                       node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)), clip_model=clip_model)
    assert model.nodes == meta["nodes"] and model.c2p == meta["c2p"]
    model.update_classifier()
    zerr = float(np.abs(model.zsl_weights.cpu().numpy() - z["zsl_weights"]).max())
    print(f"[measured] {case} zsl_weights max |HIP - reference| = {zerr:.2e}")
    assert zerr < 1e-3, zerr     # unit rows, f16 text tower: the same bar as the logits
    ev = evaluate.Evaluator(model)
    for i in range(meta["batches"]):
        img = synth.images(meta["bsz"], cfg["image_resolution"], meta["image_seed0"] + i)
        lg = model(img.to(DEV), None)
        assert lg.shape == (meta["bsz"], meta["n_nodes"])
        # north_star tolerance: fp32 logits within 1e-3 of the reference PyTorch path
        err = float(np.abs(lg.cpu().numpy() - z["logits"][i]).max())
        assert err < LOGIT_TOL[idt], f"max |logit - reference| = {err:.3e} ({idt})"
        # index work is checked bit-exactly on the REFERENCE's logits (same inputs to the kernels)
        ref_lg = torch.from_numpy(z["logits"][i]).to(DEV)
        pred, path = ev.add_batch(ref_lg, meta["targets"][i])
        assert np.array_equal(pred.cpu().numpy(), z["pred_top20"][i])
        assert np.array_equal(path.cpu().numpy().astype(np.float32), z[f"dict_path_{i}"])
    assert ev.summary() == meta["metric"]                   # the exact string the reference's main.test printed
    for k, v in meta["counters"].items():
        assert abs(ev.counters()[k] - v) < 1e-9


def test_tree_topk_indices_vs_oracle_with_margin(golden_dir, tmp_path):
    """Top-k node ids of the HIP logits equal the fp32 oracle's wherever the oracle's decision margin
    exceeds twice the measured logit error (bit-exact index parity is undecidable inside the error band)."""
    meta, z, cfg, edges = _tree_case("smallvit_n300", golden_dir)
    sd = synth.clip_state_dict(cfg, 0)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    clip_model = build_model(sd, image_dtype="f16", text_dtype="f16").to(DEV)
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"],
                       node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)), clip_model=clip_model)
    model.update_classifier()
    img = synth.images(16, cfg["image_resolution"], 4242)
    lg = model(img.to(DEV), None)
    ref = tree_ref.forward(sd, img, torch.from_numpy(z["zsl_weights"])).numpy()
    err = np.abs(lg.cpu().numpy() - ref).max()
    assert err < 1e-3
    got = ops.topk_rows(lg, 5, cols=model.test_index32).cpu().numpy()
    te = model.test_index.cpu().numpy()
    checked = 0
    for r in range(16):
        sub = ref[r, te]
        order = tree_ref.topk_desc(sub, 6)
        for j in range(5):
            if sub[order[j]] - sub[order[j + 1]] > 2 * err and (j == 0 or sub[order[j - 1]] - sub[order[j]] > 2 * err):
                assert got[r, j] == te[order[j]]
                checked += 1
    assert checked >= 40


def test_full_size_properties_vitb32():
    """BASELINE configs[1] sizes (ViT-B/32, N = 21 841): size-independent properties."""
    n, d, b = 21841, 512, 64
    z = torch.from_numpy(synth.normal(5, "z", n * d).astype(np.float32).reshape(n, d))
    z = (z / z.norm(dim=-1, keepdim=True)).to(torch.bfloat16).to(DEV)
    f = torch.from_numpy(synth.normal(6, "f", b * d).astype(np.float32).reshape(b, d))
    f = (f / f.norm(dim=-1, keepdim=True)).to(torch.bfloat16).to(DEV)
    ld = (n + 63) // 64 * 64
    lg = torch.empty(b, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f, z, lg, n=n)
    ref = f.float().cpu() @ z.float().cpu().t()
    assert (lg[:, :n].cpu() - ref).abs().max() < 2e-5       # same bf16 inputs, fp32 accumulate
    # row permutation of the class matrix permutes the logits columns
    perm = torch.from_numpy(np.argsort(synth.uniform(9, "p", n), kind="stable")).to(DEV)
    lg2 = torch.empty(b, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f, z[perm].contiguous(), lg2, n=n)
    assert torch.equal(lg2[:, :n], lg[:, :n][:, perm])
    # top-20: sorted, distinct, and really the 20 largest
    idx, val = ops.topk_rows(lg[:, :n], 20, n_cols=n, want_values=True)
    assert (val[:, :-1] >= val[:, 1:]).all()
    kth = val[:, -1:].cpu()
    assert ((lg[:, :n].cpu() > kth).sum(dim=1) <= 19).all()
    # determinism: same inputs twice -> identical bits
    lg3 = torch.empty(b, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f, z, lg3, n=n)
    assert torch.equal(lg3[:, :n], lg[:, :n])


def test_uint8_nhwc_input_matches_normalised_float_path():
    """uint8 crops normalised inside the patch kernel == ToTensor + Normalize (clip/clip.py:71-78) then the fp32 path."""
    sd = synth.clip_state_dict("small-vit", 0)
    model = build_model(sd).to(DEV)
    u8 = torch.from_numpy(synth.randint(4, "u8", 3 * 96 * 96 * 3, 0, 256).astype(np.uint8).reshape(3, 96, 96, 3))
    mean, std = torch.tensor(ops.CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(ops.CLIP_STD).view(1, 3, 1, 1)
    x = (u8.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    f_u8 = model.encode_image(u8.to(DEV)).cpu()
    f_fp = model.encode_image(x.to(DEV)).cpu()
    ref = clip_ref.encode_image(sd, x)
    assert (f_u8 - f_fp).abs().max() < 2e-3 * max(1.0, float(ref.abs().max()))
    assert (f_u8 - ref).abs().max() < 6e-3 * max(1.0, float(ref.abs().max()))


# ---- BASELINE.json configs by name ------------------------------------------------------------------------------------
def test_config0_rn50_n1000_batch32_vs_cpu_oracle(tmp_path):
    """BASELINE configs[0]: RN50 zero-shot evaluation, 1 000-class subset, batch 32 - the reference's own CPU-runnable
    case.  True-dimension RN50 (hash-seeded weights), 1 000-node hierarchy, one batch of 32: HIP logits against the
    fp32 CPU oracle within the north-star's 1e-3, top-1 equal wherever the oracle's margin exceeds the error band."""
    cfg = synth.CLIP_CONFIGS["RN50"]
    sd = synth.clip_state_dict(cfg, 0)
    n = 1000
    edges = synth.make_dag(n, 8, 7, 0.05)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], 300, 400, 13)
    tokens = synth.make_tokens(len(h.nodes), 11, vocab_size=cfg["vocab_size"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"], node_tokens=tokens, clip_model=build_model(sd).to(DEV))
    model.update_classifier()
    img = synth.images(32, 224, 1234)
    lg = model(img.to(DEV), None).cpu().numpy()
    zsl = tree_ref.update_classifier(sd, tokens, trim=True)
    zerr = float(np.abs(model.zsl_weights.float().cpu().numpy() - zsl.numpy()).max())
    print(f"[measured] RN50 N=1000 zsl_weights max |HIP - oracle| = {zerr:.2e}")
    assert zerr < 1e-3, zerr
    ref = tree_ref.forward(sd, img, zsl).numpy()
    err = float(np.abs(lg - ref).max())
    assert err < 1e-3, err
    te = model.test_index.cpu().numpy()
    got = ops.topk_rows(torch.from_numpy(lg).to(DEV), 1, cols=model.test_index32).cpu().numpy()[:, 0]
    checked = 0
    for r in range(32):
        sub = ref[r, te]
        o = tree_ref.topk_desc(sub, 2)
        if sub[o[0]] - sub[o[1]] > 2 * err:
            assert got[r] == te[o[0]]
            checked += 1
    assert checked >= 8


def test_config2_rn50_hierarchy_full_size_metrics_consistent():
    """BASELINE configs[2] sizes: N = 20 842 nodes, 7 400 seen / 13 442 unseen.  The fused evaluation kernel
    (top-20 over the unseen columns, top-1 over the seen ones, per-level arg-max) against the separate kernels on the
    same full-size logits, and the counters against a host recount."""
    n, d, b = 20842, 1024, 48
    edges = synth.make_dag(n, 12, 7, 0.03)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    nn_ = len(h.nodes)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], 7400, 13442, 13)
    idx = {w: i for i, w in enumerate(h.nodes)}
    tr = torch.tensor([idx[w] for w in splits["train"]], dtype=torch.int32, device=DEV)
    te = torch.tensor([idx[w] for w in splits["rest"]], dtype=torch.int32, device=DEV)
    assert tr.numel() == 7400 and te.numel() == 13442
    depth = torch.tensor([len(p) for p in h.c2p], dtype=torch.int32)
    z = torch.from_numpy(synth.normal(5, "z2", nn_ * d).astype(np.float32).reshape(nn_, d))
    z = (z / z.norm(dim=-1, keepdim=True)).half().to(DEV)
    f = torch.from_numpy(synth.normal(6, "f2", b * d).astype(np.float32).reshape(b, d))
    f = (f / f.norm(dim=-1, keepdim=True)).half().to(DEV)
    ld = (nn_ + 63) // 64 * 64
    lg = torch.empty(b, ld, dtype=torch.float32, device=DEV)
    ops.gemm_nt(f, z, lg, n=nn_)
    n_levels = int(depth.max()) + 1
    index = ops.EvalIndex(depth.to(DEV), tr, te, n_levels)
    lv, p1, pk = ops.eval_rows(lg, index, 20)
    assert torch.equal(pk, ops.topk_rows(lg, 20, cols=te))
    lv2, p12 = ops.level_argmax(lg, depth.to(DEV), n_levels, cols=tr, want_top1=True)
    assert torch.equal(lv, lv2) and torch.equal(p1, p12)
    # host recount of the top-1 over the seen columns and of one level
    sub = lg[:, :nn_].cpu()[:, tr.cpu().long()]
    assert torch.equal(p1.cpu()[:, 0].long(), tr.cpu().long()[sub.argmax(1)])
    lvl = 5
    mask = (depth[tr.cpu().long()] == lvl)
    if bool(mask.any()):
        cols = tr.cpu().long()[mask]
        assert torch.equal(lv.cpu()[:, lvl].long(), cols[lg[:, :nn_].cpu()[:, cols].argmax(1)])


def test_graph_replay_equals_eager_and_survives_changes(golden_dir, tmp_path):
    """forward() as a HIP graph: bit-identical to eager launches; fresh output tensors; new input buffers, a new batch
    size and a new classifier each start a clean generation; never-repeating addresses fall back to a static buffer."""
    meta, z, cfg, edges = _tree_case("smallvit_n300", golden_dir)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"], node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)),
                       clip_model=build_model(synth.clip_state_dict(cfg, 0)).to(DEV))
    model.update_classifier()
    res = cfg["image_resolution"]
    imgs = [synth.images(6, res, 50 + i).to(DEV) for i in range(3)]
    model.use_graph = False
    eager = [model(x, None).clone() for x in imgs]
    model.use_graph = True
    first = model(imgs[0], None)
    again = model(imgs[0], None)
    assert torch.equal(first, eager[0]) and torch.equal(again, eager[0]) and first.data_ptr() != again.data_ptr()
    for x, e in zip(imgs, eager):                                 # other buffers: captured on first sight, replayed after
        assert torch.equal(model(x, None), e) and torch.equal(model(x, None), e)
    x2 = synth.images(10, res, 99).to(DEV)                        # another batch size: new generation
    model.use_graph = False; e2 = model(x2, None).clone(); model.use_graph = True
    assert torch.equal(model(x2, None), e2) and torch.equal(model(imgs[1], None), eager[1])
    keep = []
    for i in range(12):                                           # a loader that never reuses an address
        keep.append(imgs[i % 3].clone())                          # all clones stay alive: 12 distinct addresses
        assert torch.equal(model(keep[-1], None), eager[i % 3])
    assert model._graph_static is not None


def test_vit_b16_image_tower_vs_oracle():
    """ViT-B/16 (197 tokens, patch 16: the third ViT geometry besides B/32's 50 and L/14's 257 tokens) against the CPU
    oracle, whose ViT path is pinned by the reference fixtures of the other two."""
    cfg = synth.CLIP_CONFIGS["ViT-B/16"]
    sd = synth.clip_state_dict(cfg, 0)
    model = build_model(sd).to(DEV)
    img = synth.images(2, 224, 31)
    got = model.encode_image(img.to(DEV)).cpu()
    ref = clip_ref.encode_image(sd, img)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) < 2e-3 * scale
    gn, rn = got / got.norm(dim=-1, keepdim=True), ref / ref.norm(dim=-1, keepdim=True)
    assert float((gn - rn).abs().max()) < 1e-3                  # what the logits see


@pytest.mark.parametrize("arch,nodes", [("ViT-B/32", 21841), ("RN50", 20842)])
def test_batch512_forward_through_graph_vs_oracle_subsample(arch, nodes, tmp_path):
    """BASELINE configs[1] / configs[2] AT THEIR SIZE through the path the headline number runs: batch 512 (256^2 / split
    tile plans, 256^2 implicit convolutions), tree_model.forward with HIP-graph replay on - against the fp32 CPU oracle
    (model/clip_tree.py:328-333; clip/model.py:219-236 / :135-150) on a 48-row subsample.
      (1) both towers from the oracle on a 2 048-column subsample: |logit - oracle| < 1e-3 (north_star tolerance);
      (2) image tower from the oracle x the model's own class matrix over ALL columns: < 1e-3, and hit@1 over the test
          columns equal on every row whose oracle margin exceeds 2 x the measured error (count asserted and printed)."""
    from hgr_net_amd.hierarchy import build_hierarchy
    cfg = synth.CLIP_CONFIGS[arch]
    sd = synth.clip_state_dict(cfg, 0)
    edges = synth.make_dag(nodes, depth=12, seed=7, multi_parent=0.03)
    h = build_hierarchy(edges)
    n_test = int(round(nodes * 13442 / 20842))
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], nodes - n_test, n_test, 13)
    tokens = synth.make_tokens(nodes, 11, cfg["vocab_size"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"], node_tokens=tokens,
                       clip_model=build_model(sd, image_dtype="f16", text_dtype="f16").to(DEV))
    if not model.use_graph:
        pytest.skip("HGR_GRAPH=0: this test is about the graph-replayed path")
    model.update_classifier()
    img = synth.images(512, cfg["image_resolution"], 4321)
    dimg = img.to(DEV)
    lg_first = model(dimg, None).clone()                 # captures (warm-up + capture + first replay)
    assert len(model._graphs) == 1
    lg = model(dimg, None)                               # pure replay of the captured graph
    assert torch.equal(lg, lg_first)
    assert torch.equal(lg, model._forward_eager(dimg))   # and the eager launches give the same bits
    # the towers really ran on the LayerNorm-folded path: a range-guard trip falls back to the fp32 stream SILENTLY apart from a warning
    # (correct results, 20 % slower - how a decode bug of the residual pair once hid behind green tests)
    assert model.clip_model._ln_off == set() and model.clip_model.ln_guard_tripped() == {}
    rows = torch.arange(5, 512, 11)[:48]
    got = lg[rows.to(DEV)].cpu()
    # (2) oracle image tower, the model's class matrix
    ref = tree_ref.forward(sd, img[rows], model.zsl_weights.float().cpu())
    err = float((got - ref).abs().max())
    assert err < 1e-3, f"{arch} batch 512: max |logit - oracle| = {err:.3e}"
    te = model.test_index.cpu()
    top2 = ref[:, te].topk(2, dim=1)
    decidable = (top2.values[:, 0] - top2.values[:, 1]) > 2 * err
    same = got[:, te].argmax(1) == top2.indices[:, 0]
    assert int(decidable.sum()) >= 24, int(decidable.sum())
    assert bool(same[decidable].all())
    # (1) both towers from the oracle on a column subsample
    cols = torch.randperm(nodes, generator=torch.Generator().manual_seed(3))[:2048]
    z_ref = tree_ref.update_classifier(sd, tokens[cols], trim=True)
    ref_full = tree_ref.forward(sd, img[rows], z_ref)
    err_full = float((got[:, cols] - ref_full).abs().max())
    assert err_full < 1e-3, f"{arch} batch 512: max |logit - full oracle| = {err_full:.3e}"
    print(f"\n[batch512 {arch}] max|logit-oracle| {err:.2e} (image oracle x HIP class matrix), {err_full:.2e} (both towers oracle, 2048 cols); "
          f"hit@1 equal {int(same.sum())}/48, decidable {int(decidable.sum())}, equal&decidable {int((same & decidable).sum())}")
    # (3) the route bench.py TIMES - Evaluator.add_images -> tree_model.forward_eval -> hgr_logits_eval (no logits written) - at this
    #     size through the real towers (main.py:136-176 consuming clip_tree.py:331): its top-20 / top-1 / per-level ids are, bit for
    #     bit, those of forward() + hgr_eval_rows, and equal the ids derived from the ORACLE's logits wherever the oracle decides
    #     by more than 2 x the measured logit error
    ev = evaluate.Evaluator(model)
    assert ev.fused_ok()
    plan = ops.LogitsEvalPlan(ev.index)
    lv_f, p1_f, pk_f = [t.clone() for t in model.forward_eval(dimg, plan, 20)]      # capture + replay
    lv_f2, p1_f2, pk_f2 = model.forward_eval(dimg, plan, 20)                        # pure replay
    assert torch.equal(lv_f, lv_f2) and torch.equal(p1_f, p1_f2) and torch.equal(pk_f, pk_f2)
    lv_u, p1_u, pk_u = ops.eval_rows(lg, ev.index, 20)
    assert torch.equal(pk_f, pk_u), "top-20 ids of the fused route differ from forward() + hgr_eval_rows"
    assert torch.equal(p1_f, p1_u) and torch.equal(lv_f, lv_u)
    tr = model.train_index.cpu()
    depth = model.depth32.cpu().long()
    pk_s, p1_s, lv_s = pk_f[rows.to(DEV)].cpu().long(), p1_f[rows.to(DEV)].cpu().long()[:, 0], lv_f[rows.to(DEV)].cpu().long()
    top21 = ref[:, te].topk(21, dim=1)
    gaps = top21.values[:, :-1] - top21.values[:, 1:]
    dec20 = (gaps > 2 * err).all(dim=1)
    assert bool((pk_s[dec20] == te[top21.indices[:, :20]][dec20]).all())
    prefix = (gaps > 2 * err).long().cumprod(dim=1)                 # ranks decided so far: the decidable prefix of every row must agree
    assert bool(((pk_s == te[top21.indices[:, :20]]) | (prefix == 0)).all())
    t2 = ref[:, tr].topk(2, dim=1)
    dec1 = (t2.values[:, 0] - t2.values[:, 1]) > 2 * err
    assert bool((p1_s[dec1] == tr[t2.indices[:, 0]][dec1]).all())
    lv_checked = 0
    for l in range(ev.n_levels):
        cols_l = tr[depth[tr] == l]
        if cols_l.numel() < 2:
            continue
        v = ref[:, cols_l].topk(2, dim=1)
        d_l = (v.values[:, 0] - v.values[:, 1]) > 2 * err
        assert bool((lv_s[d_l, l] == cols_l[v.indices[:, 0]][d_l]).all()), f"level {l}"
        lv_checked += int(d_l.sum())
    assert lv_checked >= 48
    print(f"[batch512 {arch}] fused route == forward() + eval_rows bit for bit on 512 rows; vs oracle ids: top-20 rows fully decidable "
          f"{int(dec20.sum())}/48 equal, decided rank prefixes equal on all 48, train top-1 decidable {int(dec1.sum())}/48 equal, "
          f"{lv_checked} decidable (row, level) arg-max ids equal")


@pytest.mark.parametrize("arch,batch", [("small-vit", 5), ("ViT-B/32", 24)])
def test_last_block_on_class_tokens_only_gives_the_same_bits(arch, batch):
    """The visual head reads only the class token of the last block (clip/model.py:231: ln_post(x[:, 0, :])); out_proj, ln_2 and the
    MLP are per-token (clip/model.py:186-187), so the last block runs them on the class-token rows alone (strided views of the pair,
    same kernels).  Features must equal, bit for bit, the ones computed with every token carried through the last block."""
    from hgr_net_amd.clip import model as clip_model
    cfg = synth.CLIP_CONFIGS[arch]
    m = build_model(synth.clip_state_dict(cfg, 0)).to(DEV)
    img = synth.images(batch, cfg["image_resolution"], 77).to(DEV)
    assert clip_model.CLS_LAST
    fast = m.encode_image(img).clone()
    clip_model.CLS_LAST = False
    try:
        full = m.encode_image(img).clone()
    finally:
        clip_model.CLS_LAST = True
    assert torch.isfinite(fast).all() and torch.equal(fast, full)


def test_graphs_survive_a_workspace_reallocation(golden_dir, tmp_path):
    """a direct encode_image call with a LARGER batch between two graphed forwards re-allocates the shared
    workspace; the old graph must not be replayed on the freed buffers (workspace epoch is part of the graph key)."""
    meta, z, cfg, edges = _tree_case("smallvit_n300", golden_dir)
    sd = synth.clip_state_dict(cfg, 0)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"],
                       node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)), clip_model=build_model(sd).to(DEV))
    model.update_classifier()
    img = synth.images(8, cfg["image_resolution"], 5).to(DEV)
    a = model(img, None).clone()
    ep = model.clip_model._ws.epoch
    big = synth.images(64, cfg["image_resolution"], 6).to(DEV)
    model.clip_model.encode_image(big)                   # grows every workspace buffer
    assert model.clip_model._ws.epoch > ep
    junk = [torch.full((1 << 20,), 7.0, device=DEV) for _ in range(8)]   # recycle the freed blocks
    b = model(img, None).clone()
    del junk
    assert torch.equal(a, b)
    assert torch.equal(model(img, None), a)


def test_fused_evaluation_equals_logits_then_eval(golden_dir, tmp_path):
    """Evaluator.add_images (image tower -> hgr_logits_eval, graph replay on) advances the nine counters of main.test exactly like
    add_batch(model(imgs)) (main.py:131-191), on the reference fixture's model; and evaluate.test, which takes the fused route,
    prints the metric string the reference printed (HGR logits differ from the fixture's fp32 logits by < 1e-3, so the string is
    compared with the unfused route's, which the other tests tie to the reference)."""
    meta, z, cfg, edges = _tree_case("smallvit_n300", golden_dir)
    sd = synth.clip_state_dict(cfg, 0)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"],
                       node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)), clip_model=build_model(sd).to(DEV))
    model.update_classifier()
    ev_a, ev_b = evaluate.Evaluator(model), evaluate.Evaluator(model)
    assert ev_b.fused_ok()
    for i in range(meta["batches"]):
        img = synth.images(meta["bsz"], cfg["image_resolution"], meta["image_seed0"] + i).to(DEV)
        pa = ev_a.add_batch(model(img, None), meta["targets"][i])
        pb = ev_b.add_images(img, meta["targets"][i], want_outputs=True)
        assert torch.equal(pa[0], pb[0]) and torch.equal(pa[1], pb[1])
    assert ev_a.counters() == ev_b.counters() and ev_a.summary() == ev_b.summary()

    def loader():
        for i in range(meta["batches"]):
            yield {"img": synth.images(meta["bsz"], cfg["image_resolution"], meta["image_seed0"] + i)[None],
                   "label": torch.full((1, meta["bsz"]), meta["targets"][i], dtype=torch.long)}
    out = evaluate.test(model.opts, model, DEV, None, loader=loader(), log=False)
    assert out == ev_a.summary()


def test_pipelined_evaluation_steps_equal_single_stream_steps(golden_dir, tmp_path):
    """tree_model.forward_eval_overlapped: every evaluation step as two HIP graphs (head on the caller's stream, class-token tail +
    class logits + evaluation + counters on a second stream, step parities on alternating workspace sets) must advance the counters of
    main.py:131-191 EXACTLY like the single-graph route - over an odd number of steps, with recycled and with never-repeating input
    buffers (static-input fallback), with a step that asks for its outputs in between (it joins the tail stream) and after new weights
    (new generation)."""
    from hgr_net_amd.model import clip_tree
    meta, z, cfg, edges = _tree_case("smallvit_n300", golden_dir)
    sd = synth.clip_state_dict(cfg, 0)
    from hgr_net_amd.hierarchy import build_hierarchy
    h = build_hierarchy(edges)
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], meta["n_train"], meta["n_test"], meta["split_seed"])
    model = tree_model(_opts(tmp_path, edges), splits["all"], splits["rest"],
                       node_tokens=torch.from_numpy(z["node_tokens"].astype(np.int64)), clip_model=build_model(sd).to(DEV))
    model.update_classifier()
    te = model.test_index.cpu().tolist()
    imgs = [synth.images(meta["bsz"], cfg["image_resolution"], 900 + i).to(DEV) for i in range(13)]
    tgs = [te[(5 * i) % len(te)] for i in range(13)]

    bufs = [torch.empty_like(imgs[0]) for _ in range(3)]          # a loader that recycles three input buffers

    def run(overlap: bool, fresh_buffers: bool):
        clip_tree.TAIL_OVERLAP = overlap
        ev = evaluate.Evaluator(model)
        outs = None
        for i in range(13):
            x = bufs[i % 3]
            x.copy_(imgs[i])
            if i == 6:
                outs = ev.add_images(x, tgs[i], want_outputs=True)
                outs = (outs[0].clone(), outs[1].clone())
            else:
                lab = torch.full((meta["bsz"],), tgs[i], dtype=torch.long, device=DEV) if i % 3 == 0 else None
                ev.add_images(x, tgs[i], lab)
        return ev.counters(), ev.summary(), outs

    try:
        base = run(False, False)
        assert model._pipe is None or not model._pipe["graphs"]
        got = run(True, False)
        assert model._pipe is not None and model._pipe["ok"] and len(model._pipe["graphs"]) >= 2      # the pipeline really ran
        assert not model._pipe["static"]
        assert got[0] == base[0] and got[1] == base[1] and torch.equal(got[2][0], base[2][0]) and torch.equal(got[2][1], base[2][1])
        keep = []                                                   # never-repeating addresses: hold every clone alive

        def run_fresh():
            clip_tree.TAIL_OVERLAP = True
            ev = evaluate.Evaluator(model)
            for i in range(13):
                keep.append(imgs[i].clone())
                ev.add_images(keep[-1], tgs[i])
            return ev.counters()
        base2 = evaluate.Evaluator(model)
        clip_tree.TAIL_OVERLAP = False
        for i in range(13):
            base2.add_images(imgs[i], tgs[i])
        assert run_fresh() == base2.counters()
        assert model._pipe["static"], "13 distinct input addresses must have switched to the static input buffers"
        # new weights -> new generation: graphs are dropped and re-captured, results follow the weights
        with torch.no_grad():
            model.clip_model.visual.proj.mul_(-1.0)
        a = run(True, False)
        b = run(False, False)
        assert a[0] == b[0] and a[1] == b[1] and a[0] != base[0]
    finally:
        clip_tree.TAIL_OVERLAP = os.environ.get("HGR_TAIL_OVERLAP", "1") != "0"
