"""Driver with the reference's command line (main.py:14-70) for the paths this build covers:
``--train False`` zero-shot evaluation (main.py:104-222) and the OM / hierarchical training loop
(main.py:72-101,225-267), single GPU or one process per GPU under torch.distributed.run.

Batches come from the group loaders of ``hgr_net_amd.dataset`` (the reference's dataset/imagenet_group*.py
contract: split JSON -> one single-class batch per step, device-side resize / crop / normalise), from
``--synthetic`` (hash-generated images, one class per batch like the reference's GroupBatchSampler) or from
a user-supplied loader passed to ``run(opts, loader_train, loader_test)``.
"""
from __future__ import annotations

import argparse
import gc
import json
import os

import torch

from . import synth
from .evaluate import test
from .model import tree_model
from .training import FusedAdamW
from .utils import cosine_lr


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="HGR (MI355X build)")
    p.add_argument("--exp_name", default="HGR", type=str)
    p.add_argument("--folder", default="adaptive", type=str)
    p.add_argument("--device", default=0, type=int)
    p.add_argument("--print_freq", default=1000, type=int)
    p.add_argument("--debug", default=False, action="store_true")
    p.add_argument("--test_after_train", default=False, action="store_true")
    p.add_argument("--arch", default="RN50", type=str)
    p.add_argument("--template", default="TEMPLATES_STANDARD", type=str)
    p.add_argument("--model_train", default="all", type=str)
    p.add_argument("--model_test", default="rest", type=str)
    p.add_argument("--data_train", default="train", type=str)
    p.add_argument("--data_test", default="rest", type=str)
    p.add_argument("--graph_path", default="data/process_results/graph_edges_cls.json", type=str)
    p.add_argument("--split_path", default="data/process_results/splits_for_tree.json", type=str)
    p.add_argument("--num_workers", default=12, type=int)
    p.add_argument("--batch_size", default=256, type=int)
    p.add_argument("--test_batch_size", default=512, type=int)
    p.add_argument("--k_shots", default=-1, type=int)
    p.add_argument("--serial_batches", type=eval, default=True, choices=[True, False])
    p.add_argument("--n_episodes", default=-1, type=int)
    p.add_argument("--data_split_train", default="train", type=str)
    p.add_argument("--data_split_test", default="zsl_test", type=str)
    p.add_argument("--open_eval", type=eval, default=True, choices=[True, False])
    p.add_argument("--train", default=True, type=eval, choices=[True, False])
    p.add_argument("--lr", default=3e-7, type=float)
    p.add_argument("--w_lr", default=1e-4, type=float)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--wd", default=0.0, type=float)
    p.add_argument("--warmup_length", default=0, type=int)
    p.add_argument("--num_compare", default=256, type=int)
    p.add_argument("--weights", default="adaptive", type=str)
    p.add_argument("--training_method", default="OM", type=str)
    p.add_argument("--sample_strategy", default="topk", type=str)
    p.add_argument("--k", default=1, type=int)
    p.add_argument("--out_ratio", default=0.25, type=float)
    p.add_argument("--in_ratio", default=0.5, type=float)
    p.add_argument("--weighting", default="both", type=str)
    p.add_argument("--scale", default=1.0, type=float)
    p.add_argument("--fetch", default=False, action="store_true")
    p.add_argument("--fetch_path", type=str)
    p.add_argument("--load", default=False, action="store_true")
    p.add_argument("--load_path", default="none", type=str)
    p.add_argument("--from_epoch", default=-1, type=int)
    # additions of this build
    p.add_argument("--synthetic", default=0, type=int, help="N > 0: N synthetic single-class batches per epoch instead of a dataset")
    p.add_argument("--image_dtype", default="f16", type=str)
    p.add_argument("--text_dtype", default="f16", type=str)
    p.add_argument("--train_dtype", default="bf16", type=str)
    p.add_argument("--split_file", default=None, type=str, help="class -> image paths JSON (default data/{split}_split.json)")
    p.add_argument("--data_seed", default=None, type=int, help="seed of the loaders' class / image order (all ranks must agree)")
    p.add_argument("--ref_quirks", default=False, action="store_true",
                   help="reproduce the reference's missing zero_grad() (gradients accumulate across steps, SURVEY F11-i)")
    return p


def synthetic_loader(model, class_ids, n_batches: int, batch: int, seed: int = 0, rank: int = 0, world: int = 1,
                     shard_batch: bool = False):
    """Batches shaped like the reference's group loaders: {'img': [1,B,3,R,R], 'label': [1,B]}, one class each.
    Evaluation deals whole batches round-robin over the ranks (batch i goes to rank i % world).  Training
    (``shard_batch``) gives every rank ITS rows of the same single-class global batch at every step - rank r holds rows
    [r*B, (r+1)*B) of a global batch of world*B images - so that the averaged gradient is the global batch's."""
    if shard_batch:
        base = synth.images(batch * world, model.resolution, seed)[rank * batch: (rank + 1) * batch]
        steps = range(n_batches)
    else:
        base = synth.images(batch, model.resolution, seed + 17 * rank)
        steps = range(rank, n_batches, world)
    for i in steps:
        label = class_ids[(7 * i + 3) % len(class_ids)]
        yield {"img": base.roll(i, 0)[None], "label": torch.full((1, batch), label, dtype=torch.long)}


def train(opts, epoch, model, train_loader, num_batches, optimizer, optimizer2, scheduler, device, group=None):
    """One epoch of main.train (main.py:72-101): scheduler, train_batch, clip + AdamW (fused kernels), logging."""
    torch.cuda.empty_cache()
    gc.collect()
    model._dp_group = group           # data-parallel: the distinct prompts of every step are sharded over the ranks (training.OMTrainer)
    for i, data in enumerate(train_loader):
        scheduler(i + epoch * num_batches)
        imgs, targets = data["img"][0].to(device), data["label"][0].to(device)
        if not opts.ref_quirks:
            optimizer.zero_grad()
            if optimizer2 is not None:
                optimizer2.zero_grad()                    # layer_weight.grad follows the same rule as the CLIP gradients
        if group is not None and getattr(model, "_trainer", None) is not None and model._trainer.grad_ready_hook is None:
            # bucketed RCCL all-reduce on the flat gradient buffer, overlapped: the text-tower / head buckets go out while
            # the image tower is still in its backward (the last thing train_batch does, clip_tree.py:280)
            optimizer.set_late_params(model.clip_model.visual.parameters())
            model._trainer.grad_ready_hook = lambda part: optimizer.allreduce_part(part, group)
        overlapped = group is not None and getattr(model, "_trainer", None) is not None
        loss = model.train_batch(imgs, targets, opts.training_method, opts.sample_strategy)
        if group is not None and not overlapped:          # first step: the trainer did not exist before train_batch built it
            optimizer.allreduce(group)
        if group is not None and optimizer2 is not None and model.layer_weight.grad is not None:
            # EVERY step, overlapped or not: layer_weight is not in the flat gradient buffer the hooks all-reduce, and ranks
            # that step it with their local gradient weight the CE terms differently from then on (<= 13 floats)
            import torch.distributed as dist
            dist.all_reduce(model.layer_weight.grad, op=dist.ReduceOp.SUM, group=group)
            model.layer_weight.grad.div_(dist.get_world_size(group))
        optimizer.step()                                  # clip_grad_norm_(params, 1.0) + AdamW, main.py:87-91
        if optimizer2 is not None:
            optimizer2.step()
        if i % opts.print_freq == 0:
            out = "loss: {:.2f}, {}/{}".format(loss, i, num_batches)
            print(out, flush=True)
            with open(model.save_path + "arugements.log", "a") as f:
                f.writelines(out + "\n")


def run(opts, loader_train=None, loader_test=None, group=None):
    device = "cuda:{}".format(opts.device)
    splits = json.load(open(opts.split_path, "r"))
    print("Creating models")
    model = tree_model(opts, candidates_train=splits[opts.model_train], candidates_test=splits[opts.model_test],
                       node_tokens=getattr(opts, "node_tokens", None), clip_model=getattr(opts, "clip_model", None))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    base_seed = None
    if world > 1 and group is not None:
        # one seed for the whole job (rank 0's --data_seed, else drawn there): the loaders' class order and the negative sampling of
        # the OM step (Python `random`, model/clip_tree.py:116-219) must agree on every rank - each rank holds its rows of the SAME
        # single-class batch and encodes its share of the SAME prompt list
        import random
        import torch.distributed as dist
        seed_t = torch.tensor([opts.data_seed if getattr(opts, "data_seed", None) is not None else random.getrandbits(31)], dtype=torch.int64, device=device)
        dist.broadcast(seed_t, src=0, group=group)
        base_seed = int(seed_t.item())
        if getattr(opts, "data_seed", None) is None:
            opts.data_seed = base_seed
    if opts.train:
        with open(model.save_path + "arugements.log", "a") as f:
            for k, v in vars(opts).items():
                if k not in ("node_tokens", "clip_model"):
                    f.writelines(k + " : " + str(v) + "\n")
        print("Training.")
        if opts.synthetic <= 0 and loader_train is None:          # main.py:240-243: the group-batch loaders over the split file
            print("Loading datasets")
            from .dataset import DataManager
            data = DataManager(opts=opts, split=opts.data_split_train, node_set=model.nodes, candidates=splits[opts.data_train],
                               resolution=model.resolution)
            loader_train = data.get_data_loader(device=device, rank=rank, world_size=world, workers=opts.num_workers)
        num_batches = opts.synthetic if opts.synthetic > 0 else len(loader_train)
        params = [p for name, p in model.named_parameters() if p.requires_grad and name != "layer_weight"]
        optimizer = FusedAdamW(params, lr=opts.lr, weight_decay=opts.wd, max_norm=1.0)
        optimizer2 = torch.optim.SGD([model.layer_weight], lr=opts.w_lr) if opts.weights == "adaptive" else None
        scheduler = cosine_lr(optimizer, opts.lr, opts.warmup_length, opts.epochs * num_batches)
        for epoch in range(opts.from_epoch + 1, opts.epochs):
            if base_seed is not None:
                import random
                random.seed(base_seed * 1000003 + epoch)        # identical on every rank, fresh every epoch
            ids = model.train_index.tolist()
            loader = loader_train if opts.synthetic <= 0 else synthetic_loader(model, ids, opts.synthetic, opts.batch_size, epoch, rank, world, shard_batch=True)
            train(opts, epoch, model, loader, num_batches, optimizer, optimizer2, scheduler, device, group)
            if rank == 0:
                model.save(opts, epoch)
                print("Model saved.")
            if opts.test_after_train:
                te = loader_test if opts.synthetic <= 0 else synthetic_loader(model, model.test_index.tolist(), opts.synthetic, opts.test_batch_size, 99, rank, world)
                test(opts, model, device, splits, loader=te, group=group)
    else:
        print("Direct testing.")
        te = loader_test if opts.synthetic <= 0 else synthetic_loader(model, model.test_index.tolist(), opts.synthetic, opts.test_batch_size, 99, rank, world)
        return test(opts, model, device, splits, loader=te, group=group)


def main(argv=None):
    opts = build_parser().parse_args(argv)
    group = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        opts.device = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(opts.device)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{opts.device}"))
        group = dist.group.WORLD
        if os.environ.get("HGR_COMM") == "native":              # C-ABI RCCL collectives for the blocking exchanges (hgr_net_amd.comm)
            from . import comm
            comm.init_from_torch(group)
    return run(opts, group=group)


if __name__ == "__main__":
    main()
