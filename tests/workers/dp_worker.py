"""Child process of tests/test_gpu_multirank.py: one rank of a data-parallel evaluation / OM training step.

    python dp_worker.py <eval|train|adaptive> <out.pt>  (RANK / WORLD_SIZE / MASTER_* from the environment)

With HGR_TEST_ONE_GPU=1 every rank uses cuda:0 and the gloo backend (RCCL refuses two ranks on one device; the
driver's real multi-GPU runs use RCCL).  WORLD_SIZE=1 runs the same code with no process group: the 1-rank reference
the N-rank result is compared with (hard part H7)."""
import json
import os
import random
import sys
import tempfile
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

from hgr_net_amd import evaluate, synth
from hgr_net_amd.clip.model import build_model
from hgr_net_amd.hierarchy import build_hierarchy
from hgr_net_amd.model import tree_model
from hgr_net_amd.parallel import batches_of_rank

mode, out_path = sys.argv[1], sys.argv[2]
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
one_gpu = os.environ.get("HGR_TEST_ONE_GPU") == "1"
local = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = f"cuda:{local}"
group = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if one_gpu else "nccl", **({} if one_gpu else {"device_id": torch.device(dev)}))
    group = dist.group.WORLD

if mode == "native":
    # libhgr's own RCCL communicator (include/hgr.h "Data-parallel collectives", hgr_net_amd.comm) against torch.distributed on the
    # same data: all-reduce (sum, max; fp32 / fp64 / int32), all-gather, broadcast.  One rank per DEVICE (RCCL refuses two ranks
    # on one); world 1 still goes through dlopen(librccl), ncclCommInitRank and the enum mapping of every call.
    import torch.distributed as dist
    from hgr_net_amd import comm
    if group is None:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
        group = dist.group.WORLD
    comm.init_from_torch(group)
    assert comm.active() and comm.world() == world and comm.rank() == rank
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    res = {}
    for dt in (torch.float32, torch.float64, torch.int32):
        x = (torch.randn(4099, generator=g) * 100).to(dt).to(dev)
        for op, top in ((comm.SUM, dist.ReduceOp.SUM), (comm.MAX, dist.ReduceOp.MAX)):
            a, b = x.clone(), x.clone()
            comm.allreduce(a, op)
            dist.all_reduce(b, op=top, group=group)
            torch.cuda.synchronize()
            res[f"allreduce_{dt}_{op}"] = bool(torch.equal(a, b)) if dt == torch.int32 or op == comm.MAX else bool(torch.allclose(a, b, rtol=1e-6, atol=1e-4))
    loc = torch.randn(7, 33, generator=g).to(dev).half()
    out = comm.allgather(loc)
    ref = [torch.empty_like(loc) for _ in range(world)]
    dist.all_gather(ref, loc, group=group)
    torch.cuda.synchronize()
    res["allgather"] = bool(torch.equal(out, torch.stack(ref)))
    t = torch.full((1025,), float(rank + 1), device=dev)
    comm.broadcast(t, root=world - 1)
    torch.cuda.synchronize()
    res["broadcast"] = bool((t == float(world)).all())
    comm.destroy()
    assert not comm.active()
    if rank == 0:
        torch.save(res, out_path)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)

cfg = synth.CLIP_CONFIGS["small-vit"]
sd = synth.clip_state_dict(cfg, 0)
n = 301                                                   # odd: uneven text shards under world 2
edges = synth.make_dag(n, depth=8, seed=7, multi_parent=0.05)
h = build_hierarchy(edges)
splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], 100, 150, 13)
n_ctx = 4 if mode == "trainctx" else 0                     # CoOp learnable context (BASELINE configs[4]): its gradient sums over ALL prompts
tokens = synth.make_tokens(n, 11, cfg["vocab_size"], n_ctx=n_ctx)
tmp = tempfile.mkdtemp(prefix="hgr_dp_")
gp = os.path.join(tmp, "g.json")
json.dump(edges, open(gp, "w"))
opts = types.SimpleNamespace(device=dev, folder=tmp, exp_name="HGR", weights="adaptive" if mode == "adaptive" else "equal", out_ratio=0.5, in_ratio=0.5, from_epoch=-1,
                             graph_path=gp, arch="synthetic", fetch=False, load=False, load_path="none", scale=1.0, num_compare=12, k=1,
                             sample_strategy="topk", weighting="both", train_dtype="bf16", n_ctx=n_ctx)
model = tree_model(opts, splits["all"], splits["rest"], node_tokens=tokens, clip_model=build_model(sd).to(dev))
if model.ctx is not None:                                  # the reference draws the context from torch's generator (model/CoOp.py:66-68): pin it
    import numpy as np
    c0 = torch.from_numpy(synth.normal(21, "ctx", n_ctx * cfg["transformer_width"]).astype(np.float32).reshape(n_ctx, -1)) * 0.02
    model.ctx.data.copy_(c0.to(dev))
res = cfg["image_resolution"]

if mode == "eval":
    # main.test's loop (main.py:131-191) over 7 single-class batches dealt round-robin over the ranks
    model.update_classifier(group=group)
    ev = evaluate.Evaluator(model)
    te = model.test_index.tolist()
    n_batches, bsz = 7, 16
    for i in batches_of_rank(n_batches, world, rank):
        img = synth.images(bsz, res, 500 + i).to(dev)
        target = te[(7 * i + 3) % len(te)]
        logits = model(img, None, static_output=True)
        ev.add_batch(logits, target, torch.full((bsz,), target, dtype=torch.long, device=dev), want_outputs=False)
    counters = ev.counters(group)
    summary = ev.summary(group)
    if rank == 0:
        torch.save({"counters": counters, "summary": summary, "zsl": model.zsl_weights.cpu()}, out_path)
elif mode == "adaptive":
    # hgr_net_amd.main.train itself (main.py:72-101) for THREE steps with --weights adaptive: from the second step on the CLIP
    # gradients travel through the overlapped hooks, and layer_weight.grad (not in the flat buffer) must still be averaged
    # on every step or the ranks' layer weights drift apart (round-2 advisor finding)
    from hgr_net_amd import main as drv
    from hgr_net_amd.training import FusedAdamW
    from hgr_net_amd.utils import cosine_lr
    opts.ref_quirks, opts.training_method, opts.sample_strategy, opts.print_freq = False, "OM", "topk", 10 ** 9
    params = [p for nm, p in model.named_parameters() if p.requires_grad and nm != "layer_weight"]
    opt = FusedAdamW(params, lr=1e-6, max_norm=1.0)
    opt2 = torch.optim.SGD([model.layer_weight], lr=0.5)           # a large step: un-averaged gradients would show at once
    lw0 = model.layer_weight.detach().clone()
    steps, gb = 3, 8
    random.seed(5)
    loader = drv.synthetic_loader(model, model.train_index.tolist(), steps, gb // world, seed=3, rank=rank, world=world, shard_batch=True)
    drv.train(opts, 0, model, loader, steps, opt, opt2, cosine_lr(opt, 1e-6, 0, steps), dev, group)
    torch.cuda.synchronize()
    lw = model.layer_weight.detach().clone()
    every = [lw.cpu()]
    if group is not None:
        import torch.distributed as dist
        box = [torch.empty_like(lw) for _ in range(world)]
        dist.all_gather(box, lw)
        every = [b.cpu() for b in box]
    if rank == 0:
        torch.save({"layer_weight": every, "initial": lw0.cpu()}, out_path)
else:
    # one OM step on ONE single-class global batch sharded over the ranks with identical sampling seeds (H7)
    from hgr_net_amd.training import FusedAdamW
    gb = 8
    img = synth.images(gb, res, 77)
    lo, hi = rank * gb // world, (rank + 1) * gb // world
    target = max(model.train_index.tolist(), key=lambda i: (len(model.c2p[i]), -i))
    params = [p for nm, p in model.named_parameters() if p.requires_grad and nm != "layer_weight"]
    names = [nm for nm, p in model.named_parameters() if p.requires_grad and nm != "layer_weight"]
    opt = FusedAdamW(params, lr=1e-6, max_norm=1.0)
    opt.zero_grad()
    if group is not None:
        # the product path of hgr_net_amd.main from the second step on: text-tower / head buckets are all-reduced (async) right
        # before the image tower's backward, the image tower's buckets after it
        from hgr_net_amd.training import OMTrainer
        model._trainer = OMTrainer(model, opts.train_dtype)
        model._dp_group = group                                  # prompt-parallel text tower (what hgr_net_amd.main.train and bench.py set)
        opt.set_late_params(model.clip_model.visual.parameters())
        model._trainer.grad_ready_hook = lambda part: opt.allreduce_part(part, group)
    # HGR_TEST_UNSEEDED=1: every rank seeds `random` differently (what a caller of hgr_net_amd.main got before round 5): the step
    # must then adopt the first rank's draw instead of raising (round-4 advisor finding)
    random.seed(5 + (rank if os.environ.get("HGR_TEST_UNSEEDED") == "1" else 0))
    from hgr_net_amd import ops as _ops
    _ops.PROFILE = []                                            # algorithmic FLOPs of every GEMM this rank launches in the step
    loss = model.train_batch(img[lo:hi].to(dev), torch.full((hi - lo,), target, dtype=torch.long, device=dev), "OM", "topk")
    recs, _ops.PROFILE = _ops.PROFILE, None
    gemm_flops = float(sum(r[3] for r in recs))
    loss_t = torch.tensor([loss], dtype=torch.float64, device=dev)
    if group is not None:
        import torch.distributed as dist
        dist.all_reduce(loss_t)
        loss_t /= world
    torch.cuda.synchronize()
    if rank == 0:
        grads = {nm: (p.grad.detach() * opt.grad_scale).cpu() for nm, p in zip(names, params)}
    ctx_all = None
    if model.ctx is not None:                                    # the context gradient as every rank holds it after the all-reduce
        cg = (model.ctx.grad.detach() * opt.grad_scale).contiguous()
        ctx_all = [cg.cpu()]
        if group is not None:
            box = [torch.empty_like(cg) for _ in range(world)]
            dist.all_gather(box, cg)
            ctx_all = [b.cpu() for b in box]
    contra_all = [model._trainer.last_contra]
    if group is not None:
        contra_all = [None] * world
        dist.all_gather_object(contra_all, model._trainer.last_contra)
    if rank == 0:
        uniq = len({i for ids, _ in model._trainer.last_contra for i in ids})
        torch.save({"loss": float(loss_t.item()), "grads": grads, "contra": model._trainer.last_contra, "gemm_flops": gemm_flops,
                    "contra_per_rank": contra_all,
                    "text_rows": model._trainer.last_text_rows, "uniq": uniq, "ctx_grad_per_rank": ctx_all}, out_path)
if group is not None:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
