"""Data parallelism for the zero-shot path: one process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI on ROCm).  The reference has no multi-GPU code at all; the path
shards naturally because evaluation batches are independent single-class units (F6):

* hot loop: rank r evaluates batches r, r + world, ...  - NO collective inside the loop;
* one exchange step before it: the class-embedding matrix zsl_weights [N, D].  Each rank text-encodes
  a contiguous N/world slice of the prompts and the slices are all-gathered (each rank contributes
  N/world x D fp32 = 5.6 MB at N = 21 841, D = 512, world 8), which also divides the text-tower work
  by `world`; a broadcast from rank 0 (what the north-star wording suggests) would move 8x the bytes
  and leave 7 GPUs idle during the text pass;
* one all-reduce(sum) of the 9 metric counters at the end (evaluate.Evaluator.counters).

The same functions run on CPU tensors over the "gloo" backend, which is how the sharding logic is
tested without GPUs (tests/test_abi_and_host.py::test_world_size_2_gloo_sharding; N-rank == 1-rank numerics: tests/test_gpu_multirank.py).
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def native_comm():
    """The libhgr RCCL communicator (hgr_net_amd.comm) when HGR_COMM=native selected it and it has been initialised
    (bench.py / main.py do that right after the process group exists: the group only bootstraps the unique id); else None
    and torch.distributed carries the collectives."""
    import os
    if os.environ.get("HGR_COMM") != "native":
        return None
    from . import comm
    return comm if comm.active() else None


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's rows; the first n % world ranks get one extra row."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(local: torch.Tensor, n: int, group=None) -> torch.Tensor:
    """Concatenate per-rank row slices (sizes from shard_bounds) into the full [n, D] matrix on every rank."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]
    tail = tuple(local.shape[1:])
    full = torch.empty((n,) + tail, dtype=local.dtype, device=local.device)
    nc = native_comm() if local.is_cuda else None
    if len(set(sizes)) == 1:
        if nc is not None:
            nc.allgather(local.contiguous(), full.view((world,) + tuple(local.shape)))   # hgr_allgather: RCCL through the C ABI
        elif local.is_cuda:
            dist.all_gather_into_tensor(full, local.contiguous(), group=group)   # one in-place RCCL all-gather
        else:
            dist.all_gather(list(full.chunk(world)), local.contiguous(), group=group)
        return full
    # all_gather needs equal sizes: pad every slice to the largest, gather, trim
    mx = max(sizes)
    pad = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    if nc is not None:
        buf = list(nc.allgather(pad))
    else:
        buf = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(buf, pad, group=group)
    lo = 0
    for r in range(world):
        full[lo: lo + sizes[r]] = buf[r][: sizes[r]]
        lo += sizes[r]
    return full


def sharded_rows(encode: Callable[[torch.Tensor], torch.Tensor], rows: torch.Tensor, group=None) -> torch.Tensor:
    """encode(rows[lo:hi]) on every rank, all-gathered to the full result."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(rows.shape[0], world, rank)
    return all_gather_rows(encode(rows[lo:hi]), rows.shape[0], group)


def sharded_text_features(clip_model, node_tokens: torch.Tensor, group=None, ctx=None) -> torch.Tensor:
    """Text features [N, D] with the prompt rows sharded over the ranks of `group`."""
    return sharded_rows(lambda t: clip_model.encode_text(t, ctx=ctx), node_tokens, group)


def batches_of_rank(num_batches: int, world: int, rank: int) -> range:
    """Indices of the evaluation batches rank `rank` owns (round-robin keeps class sizes balanced)."""
    return range(rank, num_batches, world)


def allreduce_grads(params, group=None, bucket_bytes: int = 64 << 20) -> None:
    """Average ``.grad`` of `params` over the ranks of `group`: the DP step of OM training.

    The reference has no distributed code; the build shards ONE single-class batch over the ranks with
    identical sampling seeds (so every rank contrasts against the same negatives); the mean-CE
    gradient of the global batch is then the average of the per-rank gradients.  Gradients are packed into
    flat fp32 buckets (default 64 MB: large enough to run RCCL's all-reduce at link rate over xGMI, small
    enough to pipeline packing / reduction / unpacking) and reduced bucket by bucket with one
    ``all_reduce(sum)`` each, then scaled by 1/world while unpacking.  Parameters that received no
    gradient on this rank contribute zeros so that every rank issues the same collectives.
    """
    world = dist.get_world_size(group)
    params = [p for p in params if p.requires_grad]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p.data)).reshape(-1).float() for p in bucket])
        nc = native_comm() if flat.is_cuda else None
        if nc is not None:
            nc.allreduce(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off: off + n].view_as(p.data)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
