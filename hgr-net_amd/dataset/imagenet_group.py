"""Group-batch input pipeline: every batch holds images of ONE class (the contract `tree_model.train_batch` and
`main.test` rely on: they read the class from `targets[0]`).

Mirrors the reference's dataset/imagenet_group.py (training, :37-184) and dataset/imagenet_group_test.py (evaluation):
same constructor arguments, the same `data/{split}_split.json` input (class id -> list of image paths), the same batch
dict {'img': [1, B, 3, R, R], 'label': [1, B], 'path': [...]}, the same sampler orders, `n_episodes` and
`loader.batch_sampler.num_batch`.  What differs is where the work runs:

  * the reference runs PIL decode + torchvision Resize / CenterCrop / ToTensor / Normalize in 12 DataLoader workers and
    ships fp32 tensors (602 KB per image) over PCIe;
  * here host threads only decode; the raw RGB bytes go up in one pinned copy per batch and `hgr_preprocess_bicubic`
    does the transform on the GPU (bit-exact with Pillow, see hgr_net_amd/preprocess.py).  `output="u8"` keeps the crop
    as uint8 NHWC for the ViT tower's fused normalise-and-patch kernel.

Data parallelism: evaluation batches are dealt round-robin to the ranks (`shard="batch"`); a training batch is split
across the ranks (`shard="within"`: rank r decodes images r, r + world, ... of every batch, all ranks walk the same
class order - pass the same `seed`), so every rank contrasts the same class against the same negatives.
"""
from __future__ import annotations

import json
import math
import random
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch

from ..preprocess import BatchPreprocessor


class GroupDataset:
    """The images of one class (imagenet_group.py:150-167): decode to RGB bytes; the transform runs on the device."""

    def __init__(self, img_paths: Sequence[str], label: int):
        self.img_paths = list(img_paths)
        self.label = int(label)

    def decode(self, i: int) -> np.ndarray:
        import PIL
        from PIL import Image
        try:
            img = Image.open(self.img_paths[i]).convert("RGB")
        except PIL.UnidentifiedImageError:                       # the reference substitutes the group's first image
            img = Image.open(self.img_paths[0]).convert("RGB")
        return np.asarray(img, dtype=np.uint8)

    def __len__(self) -> int:
        return len(self.img_paths)


class _GroupBatches:
    """What `iter(DataLoader(group_dataset, batch_size, shuffle, drop_last=False))` yields, as index lists."""

    def __init__(self, n: int, batch_size: int, shuffle: bool, rng: random.Random):
        self.n, self.batch_size, self.shuffle, self.rng = n, batch_size, shuffle, rng
        self._order: List[int] = []
        self._pos = 0
        self.restart()

    def __len__(self) -> int:
        return math.ceil(self.n / self.batch_size)

    def restart(self) -> None:
        self._order = list(range(self.n))
        if self.shuffle:
            self.rng.shuffle(self._order)
        self._pos = 0

    def next(self) -> List[int]:
        if self._pos >= self.n:
            raise StopIteration
        out = self._order[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return out


class ImageDataset:
    """One batch iterator per non-empty class; item i = the next batch of class-group i (imagenet_group.py:112-147)."""

    def __init__(self, data_grouped: Dict[str, Sequence[str]], node_set: Sequence[str], batch_size: int, serial_batches: bool,
                 shuffle: bool, rng: Optional[random.Random] = None):
        self.batch_size, self.serial_batches = batch_size, serial_batches
        rng = rng or random.Random()
        index = {name: i for i, name in enumerate(node_set)}        # node_set.index(cls_name), without the O(N) scan
        self.groups: List[GroupDataset] = []
        self.group_loaders: List[_GroupBatches] = []
        for cls_name, cls_group in data_grouped.items():
            if len(cls_group) > 0:
                self.groups.append(GroupDataset(cls_group, index[cls_name]))
                self.group_loaders.append(_GroupBatches(len(cls_group), batch_size, shuffle, rng))

    def __getitem__(self, i: int):
        """(group, indices of its next batch).  An exhausted group starts a new pass (the reference's
        `next(iter(...))` retry does this for `serial_batches`; without it the reference would stop the epoch)."""
        try:
            idx = self.group_loaders[i].next()
        except StopIteration:
            self.group_loaders[i].restart()
            idx = self.group_loaders[i].next()
        return self.groups[i], idx

    def __len__(self) -> int:
        return len(self.group_loaders)


class GroupBatchSampler:
    """Training order (imagenet_group.py:170-184): passes over a shuffled list of the groups, one group per batch.
    Like the reference it yields (n_episodes // n_groups + 1) * n_groups items while len() reports n_episodes."""

    def __init__(self, n_episodes: int, n_groups: int, rng: Optional[random.Random] = None):
        self.n_episodes, self.n_groups = n_episodes, n_groups
        self.rng = rng or random

    def __len__(self) -> int:
        return self.n_episodes

    def __iter__(self) -> Iterator[List[int]]:
        for _ in range(self.n_episodes // self.n_groups + 1):
            seq = list(range(self.n_groups))
            self.rng.shuffle(seq)
            for g in seq:
                yield [g]


class GroupBatchSamplerTest:
    """Evaluation order (imagenet_group_test.py:150-163): every batch of group 0, then of group 1, ..."""

    def __init__(self, all_loaders: Sequence[_GroupBatches]):
        self.all_loaders = all_loaders
        self.len_loader = [len(loader) for loader in all_loaders]
        self.num_batch = sum(self.len_loader)

    def __len__(self) -> int:
        return self.num_batch

    def __iter__(self) -> Iterator[List[int]]:
        for i, len_loader in enumerate(self.len_loader):
            for _ in range(len_loader):
                yield [i]


class GroupLoader:
    """The outer DataLoader of the reference: iterating yields batch dicts.  Decode runs `prefetch` batches ahead on
    `workers` threads (PIL releases the GIL while decoding); the device transform runs when the batch is handed out."""

    def __init__(self, dataset: ImageDataset, batch_sampler, resolution: int, device=None, output: str = "f32", workers: int = 8,
                 prefetch: int = 2, rank: int = 0, world_size: int = 1, shard: str = "batch", transform: bool = True):
        assert shard in ("batch", "within") and output in ("f32", "u8")
        self.dataset, self.batch_sampler = dataset, batch_sampler
        self.resolution, self.device, self.output = resolution, device, output
        self.workers, self.prefetch = max(1, workers), max(1, prefetch)
        self.rank, self.world_size, self.shard = rank, world_size, shard
        self.transform = transform
        self._pre: Optional[BatchPreprocessor] = None

    def __len__(self) -> int:
        n = len(self.batch_sampler)
        if self.shard == "batch" and self.world_size > 1:
            return len(range(self.rank, n, self.world_size))
        return n

    def _plan(self):
        """(group, indices) of every batch this rank handles, in order.  The group iterators advance for EVERY batch on
        every rank, so all ranks agree on what each batch contains."""
        for i, item in enumerate(self.batch_sampler):
            group, idx = self.dataset[item[0]]
            if self.world_size > 1:
                if self.shard == "batch":
                    if i % self.world_size != self.rank:
                        continue
                else:
                    idx = idx[self.rank::self.world_size] or idx[:1]   # fewer images than ranks: repeat the first, stay in step
            yield group, idx

    def __iter__(self):
        if self.transform and self._pre is None:
            if self.device is None:
                raise RuntimeError("GroupLoader needs the device the transform kernel runs on (there is no CPU transform)")
            self._pre = BatchPreprocessor(self.resolution, self.device)
        with ThreadPoolExecutor(self.workers) as pool:
            pending = []
            plan = self._plan()

            def submit():
                try:
                    group, idx = next(plan)
                except StopIteration:
                    return False
                pending.append((group, idx, [pool.submit(group.decode, j) for j in idx]))
                return True

            for _ in range(self.prefetch):
                if not submit():
                    break
            while pending:
                group, idx, futs = pending.pop(0)
                submit()
                images = [f.result() for f in futs]
                label = torch.full((1, len(idx)), group.label, dtype=torch.long)
                paths = [group.img_paths[j] for j in idx]
                if not images:
                    continue
                if self.transform:
                    img = self._pre(images, output=self.output)[None]
                else:
                    img = images                                     # host-only mode (sampler tests): decoded arrays
                yield {"img": img, "label": label, "path": paths}


def _read_split(opts, split: str) -> dict:
    path = getattr(opts, "split_file", None) or "data/{}_split.json".format(split)
    return json.load(open(path))


class DataManager:
    """Training loader factory (imagenet_group.py:37-109)."""

    def __init__(self, opts, split, node_set, candidates=None, resolution=224):
        self.opts, self.split, self.node_set, self.resolution = opts, split, node_set, resolution
        self.candidates = self.node_set if candidates is None else candidates
        self.batch_size = opts.batch_size
        self.serial_batches = opts.serial_batches
        self.k_shots = opts.k_shots
        self.rng = random.Random(getattr(opts, "data_seed", None)) if getattr(opts, "data_seed", None) is not None else random.Random()
        self.data_grouped = self.read_data()
        self.num_data = sum(len(group) for group in self.data_grouped.values())
        self.n_episodes = opts.n_episodes if opts.n_episodes > 0 else self.num_data // self.batch_size + 1

    def read_data(self):
        data_grouped = defaultdict(list)
        data = _read_split(self.opts, self.split)
        num_items = num_classes = 0
        for cls in self.candidates:
            data_grouped[cls] = data[cls]
            num_items += len(data[cls])
            num_classes += 1
        print("Done reading data, number of classes: {}, images: {}".format(num_classes, num_items))
        if self.k_shots > 0:
            unseen = set(json.load(open(getattr(self.opts, "split_path", "/data/process_results/splits_for_tree.json")))["rest"])
            seen_items = unseen_items = 0
            for cls_label, cls_group in data_grouped.items():
                if cls_label in unseen:
                    if len(cls_group) > self.k_shots:
                        data_grouped[cls_label] = self.rng.sample(list(cls_group), self.k_shots)
                    unseen_items += len(data_grouped[cls_label])
                else:
                    seen_items += len(cls_group)
            print("Done preparing {}-shot datasets, number of seen images: {}, number of unseen images: {}".format(
                self.k_shots, seen_items, unseen_items))
        return data_grouped

    def get_data_loader(self, device=None, output="f32", rank=0, world_size=1, transform=True, workers=8) -> GroupLoader:
        dataset = ImageDataset(self.data_grouped, self.node_set, self.batch_size, self.serial_batches, shuffle=True, rng=self.rng)
        sampler = GroupBatchSampler(self.n_episodes, len(dataset), rng=self.rng)
        return GroupLoader(dataset, sampler, self.resolution, _device_of(self.opts, device), output, workers=workers, rank=rank,
                           world_size=world_size, shard="within", transform=transform)


class DataManager_test:
    """Evaluation loader factory (imagenet_group_test.py:40-92): sequential, every image exactly once."""

    def __init__(self, opts, split, node_set, candidates=None, resolution=224):
        self.opts, self.split, self.node_set, self.resolution = opts, split, node_set, resolution
        self.candidates = self.node_set if candidates is None else candidates
        self.batch_size = opts.test_batch_size
        self.serial_batches = True
        self.data_grouped = self.read_data()
        self.num_data = sum(len(group) for group in self.data_grouped.values())

    def read_data(self):
        data_grouped = defaultdict(list)
        data = _read_split(self.opts, self.split)
        num_items = num_classes = 0
        for cls in self.candidates:
            data_grouped[cls] = data[cls]
            num_items += len(data[cls])
            if len(data[cls]) > 0:
                num_classes += 1
        print("Done reading data, number of classes: {}, images: {}".format(num_classes, num_items))
        return data_grouped

    def get_data_loader(self, device=None, output="f32", rank=0, world_size=1, transform=True, workers=8) -> GroupLoader:
        dataset = ImageDataset(self.data_grouped, self.node_set, self.batch_size, self.serial_batches, shuffle=False)
        sampler = GroupBatchSamplerTest(dataset.group_loaders)
        return GroupLoader(dataset, sampler, self.resolution, _device_of(self.opts, device), output, workers=workers, rank=rank,
                           world_size=world_size, shard="batch", transform=transform)


def _device_of(opts, device):
    if device is not None:
        return device
    d = getattr(opts, "device", None)
    if d is None or d == "cpu":
        return None
    return "cuda:{}".format(d) if isinstance(d, int) else d
