"""GPU: the device-side image transform (hgr_preprocess_bicubic) against the Pillow / torch golden vectors and the oracle.
Integer pipeline -> equality; the fp32 normalisation is two correctly rounded operations per step -> equality too."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hgr_net_amd import preprocess
from oracle import resample_ref

GOLD = np.load(Path(__file__).parent / "golden" / "preproc.npz")


@pytest.mark.parametrize("i", range(int(GOLD["n_cases"])))
def test_golden_cases_bit_exact(i):
    n = int(GOLD[f"npx_{i}"])
    pre = preprocess.BatchPreprocessor(n, "cuda")
    u8 = pre([GOLD[f"in_{i}"]], output="u8")
    assert np.array_equal(u8[0].cpu().numpy(), GOLD[f"u8_{i}"])
    f32 = pre([GOLD[f"in_{i}"]], output="f32")
    assert np.array_equal(f32[0].cpu().numpy(), GOLD[f"f32_{i}"])


def test_mixed_size_batch_matches_oracle():
    """One launch over images of different sizes and orientations (tables padded to the batch maximum), at the real
    crop size, including a no-op resize, an upscale and a 5x downscale."""
    rng = np.random.default_rng(5)
    shapes = [(375, 500), (500, 375), (224, 224), (300, 224), (224, 601), (120, 90), (1100, 1300), (333, 499), (256, 256)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in shapes]
    imgs[1][:100, :100] = 255                       # saturated block next to noise: bicubic overshoot must clip
    imgs[1][100:200, :100] = 0
    pre = preprocess.BatchPreprocessor(224, "cuda")
    got = pre(imgs, output="u8").cpu().numpy()
    got32 = pre(imgs, output="f32").cpu().numpy()
    for k, im in enumerate(imgs):
        ref = resample_ref.transform_u8(im, 224)
        assert np.array_equal(got[k], ref), f"image {k} {im.shape}"
        assert np.array_equal(got32[k], resample_ref.normalize(ref))


def test_repeat_and_order_independent():
    """The same image gives the same bytes whatever its batch neighbours are (different table padding)."""
    rng = np.random.default_rng(6)
    a = rng.integers(0, 256, (180, 240, 3), dtype=np.uint8)
    big = rng.integers(0, 256, (900, 700, 3), dtype=np.uint8)
    pre = preprocess.BatchPreprocessor(64, "cuda")
    alone = pre([a], output="u8")[0].cpu()
    mixed = pre([big, a, big], output="u8")[1].cpu()
    assert torch.equal(alone, mixed)


def test_rejects_non_rgb_and_oversize():
    pre = preprocess.BatchPreprocessor(32, "cuda")
    with pytest.raises(ValueError):
        pre([np.zeros((40, 40), np.uint8)])
    with pytest.raises(ValueError):
        pre([np.zeros((2000, 2000, 3), np.uint8)])


def test_eval_loader_tensors_match_reference(tmp_path):
    """DataManager_test end to end (PNG files -> decode threads -> device transform) against the tensors the
    reference's DataManager_test + torchvision-equivalent transform produced for the same files."""
    import json
    import types
    from PIL import Image
    from hgr_net_amd import dataset as ds
    gold = Path(__file__).parent / "golden"
    meta = json.load(open(gold / "loader.json"))
    imgs = np.load(gold / "loader_imgs.npz")
    split = {}
    for cls, names in meta["classes"].items():
        split[cls] = []
        for name in names:
            Image.fromarray(imgs["src_" + name]).save(tmp_path / name)
            split[cls].append(str(tmp_path / name))
    json.dump(split, open(tmp_path / "val_split.json", "w"))
    t = meta["test"]
    opts = types.SimpleNamespace(split_file=str(tmp_path / "val_split.json"), test_batch_size=t["batch_size"], device=0)
    loader = ds.DataManager_test(opts, "val", t["node_set"], candidates=t["candidates"], resolution=meta["n_px"]).get_data_loader()
    n = 0
    for i, b in enumerate(loader):
        ref = imgs[f"batch_{i}"]
        assert b["img"].is_cuda and tuple(b["img"].shape) == ref.shape == tuple(t["batches"][i]["img_shape"])
        assert np.array_equal(b["img"].cpu().numpy(), ref)
        assert b["label"][0].tolist() == t["batches"][i]["label"]
        n += 1
    assert n == t["num_batch"]
    # uint8 NHWC output for the fused ViT path carries the same pixels
    u8 = next(iter(ds.DataManager_test(opts, "val", t["node_set"], candidates=t["candidates"], resolution=meta["n_px"]).get_data_loader(output="u8")))
    assert u8["img"].dtype == torch.uint8 and tuple(u8["img"].shape) == (1, 2, meta["n_px"], meta["n_px"], 3)
