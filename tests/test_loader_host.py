"""CPU: the group-batch input pipeline's host logic against what the reference's loaders did (tests/golden/loader.json,
produced by tools/make_golden_loader.py running dataset/imagenet_group*.py)."""
import json
import random
import types
from pathlib import Path

import numpy as np
import pytest

from hgr_net_amd import dataset as ds

GOLD = Path(__file__).parent / "golden"
META = json.load(open(GOLD / "loader.json"))
IMGS = np.load(GOLD / "loader_imgs.npz")


@pytest.fixture()
def tiny_dataset(tmp_path):
    """The fixture's images written back as lossless PNG files + the split file the loaders read."""
    from PIL import Image
    split = {}
    for cls, names in META["classes"].items():
        split[cls] = []
        for name in names:
            Image.fromarray(IMGS["src_" + name]).save(tmp_path / name)
            split[cls].append(str(tmp_path / name))
    path = tmp_path / "val_split.json"
    json.dump(split, open(path, "w"))
    return path


def _opts(split_file, **kw):
    base = dict(split_file=str(split_file), test_batch_size=2, batch_size=2, serial_batches=False, k_shots=0, n_episodes=0,
                device="cpu")
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_train_sampler_order_matches_reference():
    for case in META["train_sampler"]:
        s = ds.GroupBatchSampler(case["n_episodes"], case["n_groups"], rng=random.Random(case["seed"]))
        assert [g[0] for g in s] == case["order"]
        assert len(s) == case["len"]
        random.seed(case["seed"])                              # default rng = the global one, like the reference
        assert [g[0] for g in ds.GroupBatchSampler(case["n_episodes"], case["n_groups"])] == case["order"]


def test_eval_loader_batches_match_reference(tiny_dataset):
    t = META["test"]
    dm = ds.DataManager_test(_opts(tiny_dataset, test_batch_size=t["batch_size"]), "val", t["node_set"], candidates=t["candidates"],
                             resolution=META["n_px"])
    assert dm.num_data == t["num_data"]
    loader = dm.get_data_loader(transform=False)
    assert loader.batch_sampler.num_batch == t["num_batch"] == len(loader)
    got = list(loader)
    assert len(got) == len(t["batches"])
    for b, ref in zip(got, t["batches"]):
        assert b["label"].shape == (1, len(ref["label"])) and b["label"][0].tolist() == ref["label"]
        assert [Path(p).name for p in b["path"]] == ref["paths"]
        for im, name in zip(b["img"], ref["paths"]):
            assert np.array_equal(im, IMGS["src_" + name])


def test_eval_loader_round_robin_over_ranks(tiny_dataset):
    t = META["test"]
    o = _opts(tiny_dataset, test_batch_size=t["batch_size"])
    mk = lambda r, w: ds.DataManager_test(o, "val", t["node_set"], candidates=t["candidates"], resolution=16).get_data_loader(
        transform=False, rank=r, world_size=w)
    full = [[Path(p).name for p in b["path"]] for b in mk(0, 1)]
    parts = [[[Path(p).name for p in b["path"]] for b in mk(r, 3)] for r in range(3)]
    assert [len(p) for p in parts] == [len(mk(r, 3)) for r in range(3)]
    for r in range(3):
        assert parts[r] == full[r::3]


def test_train_loader_single_class_batches_and_episodes(tiny_dataset):
    node_set = META["test"]["node_set"]
    o = _opts(tiny_dataset, batch_size=2, data_seed=3)
    dm = ds.DataManager(o, "val", node_set, candidates=["n001", "n002", "n003", "n004"], resolution=16)
    assert dm.num_data == 10 and dm.n_episodes == 10 // 2 + 1
    loader = dm.get_data_loader(transform=False)
    assert len(loader.dataset) == 3                              # the empty class has no group
    batches = list(loader)
    assert len(batches) == (dm.n_episodes // 3 + 1) * 3          # the reference's sampler overshoots n_episodes the same way
    seen = {}
    for b in batches:
        labels = set(b["label"][0].tolist())
        assert len(labels) == 1 and 1 <= b["label"].shape[1] <= 2
        seen.setdefault(labels.pop(), []).extend(Path(p).name for p in b["path"])
    assert set(seen) == {1, 3, 4}
    # a group is walked without repetition until it is exhausted, then restarts
    assert sorted(seen[1][:5]) == sorted(META["classes"]["n001"])
    assert dm.n_episodes == ds.DataManager(_opts(tiny_dataset, n_episodes=4), "val", node_set, candidates=["n001"]).n_episodes + 2


def test_train_loader_splits_each_batch_over_ranks(tiny_dataset):
    node_set = META["test"]["node_set"]
    mk = lambda r, w: ds.DataManager(_opts(tiny_dataset, batch_size=4, data_seed=11), "val", node_set, candidates=["n001", "n004"],
                                     resolution=16).get_data_loader(transform=False, rank=r, world_size=w)
    full = [([Path(p).name for p in b["path"]], b["label"][0, 0].item()) for b in mk(0, 1)]
    r0 = [([Path(p).name for p in b["path"]], b["label"][0, 0].item()) for b in mk(0, 2)]
    r1 = [([Path(p).name for p in b["path"]], b["label"][0, 0].item()) for b in mk(1, 2)]
    assert len(full) == len(r0) == len(r1)                      # every rank takes part in every step
    for (pf, lf), (p0, l0), (p1, l1) in zip(full, r0, r1):
        assert lf == l0 == l1                                   # same class on every rank
        assert p0 == pf[0::2] and (p1 == pf[1::2] or (len(pf) == 1 and p1 == pf[:1]))


def test_k_shots_subsamples_unseen_classes(tiny_dataset, tmp_path):
    splits = tmp_path / "splits.json"
    json.dump({"rest": ["n001"]}, open(splits, "w"))
    o = _opts(tiny_dataset, k_shots=2, split_path=str(splits), data_seed=1)
    dm = ds.DataManager(o, "val", META["test"]["node_set"], candidates=["n001", "n004"], resolution=16)
    assert len(dm.data_grouped["n001"]) == 2 and len(dm.data_grouped["n004"]) == 3


def test_loader_without_device_refuses_to_transform(tiny_dataset):
    dm = ds.DataManager_test(_opts(tiny_dataset), "val", META["test"]["node_set"], candidates=["n001"], resolution=16)
    with pytest.raises(RuntimeError):
        next(iter(dm.get_data_loader()))


def test_clip_load_preprocess_callable_matches_oracle_transform():
    """The host `preprocess` callable returned by clip.load (the reference's per-image API) = the oracle transform."""
    from PIL import Image
    from hgr_net_amd.clip.clip import _transform
    from oracle import resample_ref
    rng = np.random.default_rng(8)
    for h, w, n in [(53, 37, 16), (40, 64, 32), (32, 32, 32), (33, 47, 32)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = _transform(n)(Image.fromarray(a)).numpy()
        assert np.array_equal(got, resample_ref.transform(a, n)), (h, w, n)
