"""GPU: CLIP towers, tree_model forward and the evaluation metrics through the C ABI, against the
fixtures captured from the reference (tests/golden) and the CPU oracle."""
import json
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
@pytest.mark.parametrize("case", ["tiny-vit", "small-vit", "ViT-B_32", "ViT-L_14", "tiny-rn", "small-rn", "RN50"])
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
    assert cos.min() > (0.9995 if dt == "bf16" else 0.99999)


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
    assert np.abs(model.zsl_weights.cpu().numpy() - z["zsl_weights"]).max() < 2e-3     # unit rows, f16 text tower
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
