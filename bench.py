#!/usr/bin/env python3
"""bench.py - images/sec of the HGR-Net zero-shot evaluation step on MI355X.

Workload (BASELINE.json configs[1], metric "images/sec over 21K-class hierarchy, batch 512"):
ViT-B/32 CLIP, N = 21 841 hierarchy nodes, batch 512 synthetic 224x224 images per GPU.  One step =
the body of the reference's hot loop (main.py:131-191): ``logits = model(imgs)`` (image tower -> L2
norm -> [512x512].[512x21841] logits GEMM, model/clip_tree.py:328-333) followed by the top-20 /
top-1 / per-level arg-max metrics.  ``update_classifier`` (text tower over all prompts) is one-off per
evaluation and runs before the timed region, sharded over ranks + RCCL all-gather when N > 1.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: spawns the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) - see DESIGN.md "Measurement" for how each field is obtained.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

N_NODES, BATCH, ARCH = 21841, 512, "ViT-B/32"
PEAK_TFLOPS_BF16 = 2500.0        # dense bf16 MFMA peak, MI355X_MICROARCH.md


def launch_ranks(argv, n: int) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILD processes
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`), relay rank 0's single
    JSON line, return non-zero if any rank failed.  This parent never touches the GPU (it does not even import torch):
    replacing or re-executing a process that has initialised HIP is not allowed on this pool."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)          # stderr passes through
    lines = []
    for line in p.stdout:
        if line.startswith("{"):
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = p.wait()
    if rc == 0 and len(lines) != 1:
        sys.stderr.write(f"[bench] expected one JSON line from rank 0, got {len(lines)}\n")
        rc = 1
    if lines:
        sys.stdout.write(lines[-1])
        sys.stdout.flush()
    return rc


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def _median_rate(fn, bs: int, repeats: int, budget_s: float):
    """images/sec from the MEDIAN of `repeats` timed calls of fn (after one untimed call); fewer repeats only if one call
    alone exceeds the budget share."""
    t0 = time.time()
    out = fn()
    first = time.time() - t0
    reps = max(3, min(repeats, int(budget_s / max(first, 1e-3))))
    ts = []
    for _ in range(reps):
        t0 = time.time()
        out = fn()
        ts.append(time.time() - t0)
    ts.sort()
    return bs / ts[len(ts) // 2], reps, out, (bs / ts[-1], bs / ts[0])


def cpu_baseline(sd, zsl_cpu, arch="ViT-B/32"):
    """The oracle (CPU fp32 restatement of the reference path, oracle/) timed on this box's host cores on a bounded
    sample of the same workload: batches of 32 images through the same tower + N-class logits + top-20; the value is
    the median of >= 5 repeats, `cores` = torch.get_num_threads() actually used.  Beside it the SURVEY 8(d) C1 line
    (RN50, N = 1 000, batch 32, fp32: BASELINE configs[0], the reference's own CPU-runnable case)."""
    import torch
    from hgr_net_amd import synth
    from oracle import tree_ref
    bs = 32
    pool = synth.images(bs * 8, 224, 99)     # every timed call takes another slice: all of them feed the parity check below
    img = pool[:bs]
    calls = []

    def run():
        i = len(calls) % 8
        lg = tree_ref.forward(sd, pool[i * bs:(i + 1) * bs], zsl_cpu)
        lg.topk(20, dim=1)
        calls.append((i, lg))
        return lg

    rate, reps, lg, (lo, hi) = _median_rate(run, bs, 7, 14.0)
    seen = dict(calls)
    idx = sorted(seen)
    # the oracle's logits of every slice it was timed on: checked against the HIP path below (one HIP batch of all of them)
    cpu_baseline.last = (torch.cat([pool[i * bs:(i + 1) * bs] for i in idx]), torch.cat([seen[i] for i in idx]))
    out = {"value": round(rate, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"median of {reps} x batch {bs} of the same {arch} N={zsl_cpu.shape[0]} forward+top20 (min {lo:.1f}, max {hi:.1f}), "
                     f"oracle/ (torch fp32 CPU, torch.get_num_threads()={torch.get_num_threads()}), host cpu_count={os.cpu_count()}"}
    # C1: RN50, 1 000 classes, batch 32 (update_classifier excluded: the class matrix is a seeded unit-norm stand-in, the
    # timed arithmetic - tower, L2 norm, [32x1024].[1024x1000], top-20 - does not depend on its values)
    try:
        cfg = synth.CLIP_CONFIGS["RN50"]
        sd_rn = synth.clip_state_dict(cfg, 0)
        g = torch.Generator().manual_seed(5)
        z = torch.randn(1000, cfg["embed_dim"], generator=g)
        z = z / z.norm(dim=1, keepdim=True)

        def run_c1():
            lgc = tree_ref.forward(sd_rn, img, z)
            lgc.topk(20, dim=1)
            return lgc

        r1, n1, _, (lo1, hi1) = _median_rate(run_c1, bs, 5, 12.0)
        out["c1"] = {"value": round(r1, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
                     "sample": f"BASELINE configs[0]: RN50 N=1000 batch {bs} fp32 forward+top20, median of {n1} (min {lo1:.1f}, max {hi1:.1f})"}
    except Exception as e:  # noqa: BLE001 - the C1 line is informative; the headline sample above is the contract
        out["c1"] = {"error": repr(e)}
    return out


def tower_forward_flops(cfg, n_img: int, n_txt: int, l_txt: int) -> float:
    """Algorithmic forward FLOPs (2 x MAC, attention included) of `n_img` images and `n_txt` prompts of `l_txt` tokens:
    per token and layer 24 W^2 (qkv, out, fc, proj) + 4 L W (q k^T and p v), plus patch embedding and the projections
    (ModifiedResNet towers: the survey's measured per-image figure)."""
    wt, d = cfg["transformer_width"], cfg["embed_dim"]
    txt = n_txt * (l_txt * cfg["transformer_layers"] * (24.0 * wt * wt + 4.0 * l_txt * wt) + 2.0 * wt * d)
    if cfg["vision_patch_size"]:
        w, ps = cfg["vision_width"], cfg["vision_patch_size"]
        g = cfg["image_resolution"] // ps
        L = g * g + 1
        img = n_img * (L * cfg["vision_layers"] * (24.0 * w * w + 4.0 * L * w) + 2.0 * g * g * 3 * ps * ps * w + 2.0 * w * d)
    else:
        img = n_img * {"RN50": 12.22e9, "RN101": 19.6e9}.get("RN50" if cfg["vision_layers"] == (3, 4, 6, 3) else "RN101", 12.22e9)
    return img + txt


def train_mode(a, model, cfg, h, rank, world, group, ranks_seen, dev, real_stdout):
    """`--mode train`: one step = one OM training step of the reference's loop (main.py:79-94 around
    model/clip_tree.py:222-281): zero_grad, train_batch('OM', 'topk') on ONE single-class batch sharded over the ranks
    (SURVEY H7), gradient all-reduce, clip_grad_norm_(1.0) + AdamW as fused kernels.  BASELINE configs[4] =
    `--arch ViT-L/14 --n-ctx 16 --train-dtype bf16`.  Reports images/sec and the achieved rate over 3 x the forward FLOPs
    the step executes (each distinct prompt of the K x M inner steps is encoded once: identical arithmetic, fewer passes)."""
    import random
    from hgr_net_amd import synth
    from hgr_net_amd.training import FusedAdamW
    params = [p for n, p in model.named_parameters() if p.requires_grad and n != "layer_weight"]
    opt = FusedAdamW(params, lr=3e-7, weight_decay=0.0, max_norm=1.0)
    img = synth.images(a.batch * world, cfg["image_resolution"], 1234)[rank * a.batch: (rank + 1) * a.batch].to(dev)
    target = max(model.train_index.tolist(), key=lambda i: (len(model.c2p[i]), -i))     # a deepest class: the longest OM schedule
    tg = torch.full((a.batch,), target, dtype=torch.long, device=dev)
    random.seed(0)                                                                     # identical negatives on every rank

    def step():
        opt.zero_grad()
        tr = getattr(model, "_trainer", None)
        if group is not None and tr is not None and tr.grad_ready_hook is None:
            # overlapped: text-tower / head buckets are all-reduced while the image tower back-propagates
            opt.set_late_params(model.clip_model.visual.parameters())
            tr.grad_ready_hook = lambda part: opt.allreduce_part(part, group)
        overlapped = group is not None and tr is not None
        loss = model.train_batch(img, tg, "OM", "topk")
        if group is not None and not overlapped:               # first step only: the trainer is built inside train_batch
            opt.allreduce(group)
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    loss0 = None
    for _ in range(a.warmup):
        loss0 = step()
    fence()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    picks = model._trainer.last_contra
    uniq = len({i for ids, _ in picks for i in ids})
    l_txt = int(model.node_tokens[:, :].argmax(dim=-1).max().item()) + 1
    fl = 3.0 * tower_forward_flops(cfg, a.batch, uniq, l_txt)
    ms = elapsed / a.steps * 1e3
    if rank == 0:
        line = {"metric": "images/sec, OM training step (model/clip_tree.py:222-281 + clip/AdamW, main.py:79-94)",
                "value": round(a.batch * world * a.steps / elapsed, 1), "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": a.train_dtype, "data": "synthetic",
                "config": {"workload": f"{a.arch} OM training step, {a.n_ctx} CoOp context vectors, N={a.nodes} nodes, depth-{len(model.c2p[target])} class: "
                                       f"{len(picks)} inner steps x <= 257 prompts ({uniq} distinct, {l_txt} tokens), batch {a.batch}/GPU",
                           "global_batch": a.batch * world, "parallelism": f"dp{world}", "weights": "random-init (hash-seeded)"},
                "roofline": {"kernel": "whole step (forward + backward GEMMs of both towers)", "bound": "mfma", "achieved": round(fl / (ms * 1e-3) / 1e12, 1),
                             "peak": PEAK_TFLOPS_BF16, "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_TFLOPS_BF16, 4), "traffic": None,
                             "flops_per_step": fl, "note": "3 x forward FLOPs of the executed passes (per rank)"},
                "cpu_baseline": None, "loss_first": loss0, "loss_last": loss,
                "peak_memory_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    global np, torch                                  # imported after the launcher decision (the parent never loads torch)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--nodes", type=int, default=N_NODES)
    ap.add_argument("--arch", default=ARCH)
    ap.add_argument("--image-dtype", default="f16")
    ap.add_argument("--text-dtype", default="f16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", dest="pcie", action="store_false", help="skip the PCIe-inclusive (host uint8 input) measurement")
    ap.add_argument("--unfused-eval", action="store_true", help="materialise the logits and run hgr_eval_rows on them (the first build's path)")
    ap.add_argument("--mode", default="eval", choices=["eval", "train"], help="eval: the BASELINE metric (zero-shot step); "
                    "train: one OM training step per step (configs[4] shape with --arch ViT-L/14 --n-ctx 16)")
    ap.add_argument("--n-ctx", type=int, default=0, help="train mode: CoOp learnable context vectors")
    ap.add_argument("--train-dtype", default="bf16")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(sys.argv[1:], a.gpus))        # parent: no GPU call before or after this point

    import numpy as np
    import torch

    # stdout carries exactly ONE line (the JSON): libraries that print banners to fd 1 (RCCL's version banner does)
    # are sent to stderr for the whole run, the result is written to the saved original descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    # HGR_TEST_ONE_GPU=1 (development only): every rank on cuda:0 with the gloo backend, so the multi-rank control flow
    # (sharded text encode + all-gather, batch dealing, counter all-reduce, max-over-ranks timing) can be exercised on a
    # single-GPU box; RCCL itself refuses two ranks on one device.  Never set by the driver.
    one_gpu = os.environ.get("HGR_TEST_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    group = None
    if world > 1 or os.environ.get("HGR_FORCE_DIST") == "1":              # world 1 under torchrun: exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))     # "nccl" is RCCL on ROCm
        group = dist.group.WORLD
    ranks_seen = 1
    if group is not None:
        import torch.distributed as dist
        ranks_seen = dist.get_world_size()                     # what the communicator (RCCL) itself reports

    from hgr_net_amd import evaluate, ops, synth
    from hgr_net_amd.clip.model import build_model
    from hgr_net_amd.hierarchy import build_hierarchy
    from hgr_net_amd.model import tree_model

    t0 = time.time()
    cfg = synth.CLIP_CONFIGS[a.arch]
    sd = synth.clip_state_dict(cfg, 0)
    edges = synth.make_dag(a.nodes, depth=12, seed=7, multi_parent=0.03)
    h = build_hierarchy(edges)
    n_test = int(round(a.nodes * 13442 / 20842))                 # seen/unseen proportion of BASELINE configs[2]
    splits = synth.make_splits(h.nodes, [len(c) == 0 for c in h.p2c], a.nodes - n_test, n_test, 13)
    tokens = synth.make_tokens(a.nodes, 11, cfg["vocab_size"], n_ctx=a.n_ctx if a.mode == "train" else 0)
    tmp = tempfile.mkdtemp(prefix="hgr_bench_")
    gp = os.path.join(tmp, "graph.json")
    json.dump(edges, open(gp, "w"))
    opts = types.SimpleNamespace(device=dev, folder=tmp, exp_name="HGR", weights="equal", out_ratio=0.25, in_ratio=0.5,
                                 from_epoch=-1, graph_path=gp, arch=a.arch, fetch=False, load=False, load_path="none", scale=1.0,
                                 num_compare=256, k=1, sample_strategy="topk", weighting="both", train_dtype=a.train_dtype,
                                 n_ctx=a.n_ctx if a.mode == "train" else 0)
    clip_model = build_model(sd, image_dtype=a.image_dtype, text_dtype=a.text_dtype).to(dev)
    model = tree_model(opts, splits["all"], splits["rest"], node_tokens=tokens, clip_model=clip_model)
    if a.mode == "train":
        train_mode(a, model, cfg, h, rank, world, group, ranks_seen, dev, real_stdout)
        return
    # synthetic single-class batches (every batch is one group), resident in HBM
    base = synth.images(a.batch, cfg["image_resolution"], 1234 + rank).to(dev)
    batches = [base, base.flip(0).contiguous()]
    te = model.test_index.cpu().tolist()
    targets = [te[(7 * i + rank) % len(te)] for i in range(a.steps + a.warmup)]
    log(f"[bench] setup {time.time() - t0:.1f}s: {a.arch}, N={a.nodes} nodes (depth {h.max_depth}), batch {a.batch}/GPU, world {world}")

    t0 = time.time()
    model.update_classifier(group=group)
    torch.cuda.synchronize()
    log(f"[bench] update_classifier (text tower, {a.nodes} prompts, one-off, untimed): {time.time() - t0:.2f}s")

    ev = evaluate.Evaluator(model)
    # the evaluation loop never looks at the logits themselves: the class-logits GEMM runs with the top-20 / top-1 / per-level
    # arg-max consumers in its epilogue (hgr_logits_eval), nothing [B, N] is written (--unfused-eval: logits + hgr_eval_rows)
    fused_eval = ev.fused_ok() and not a.unfused_eval

    def step(i):
        if fused_eval:
            ev.add_images(batches[i & 1], targets[i])
            return
        logits = model(batches[i & 1], None, static_output=True)      # consumed by add_batch before the next forward
        ev.add_batch(logits, targets[i], want_outputs=False)

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / a.steps * 1e3
    value = a.batch * world * a.steps / elapsed
    summary = ev.summary(group)

    # roofline of the dominant kernel (gemm_nt_128, the tower GEMMs): algorithmic FLOPs of every launch
    # (2*M*N*K) / that launch's duration, from HIP events recorded on the launch stream around each call
    # in a second, instrumented pass over the same steps (events perturb the timed pass, so it is separate).
    roof = None
    if rank == 0:
        ops.PROFILE = []
        graph_mode, model.use_graph = model.use_graph, False     # per-launch events need eager launches (a replayed graph has no host calls)
        for i in range(min(a.steps, 5)):
            step(a.warmup + i)
        torch.cuda.synchronize()
        model.use_graph = graph_mode
        recs, ops.PROFILE = ops.PROFILE, None
        tower = [(s_.elapsed_time(e_) * 1e-3, fl, by) for (name, s_, e_, fl, by, tag) in recs if name == "gemm_nt" and tag != "logits" and fl > 2e10]
        if tower:
            tsum = sum(t for t, _, _ in tower)
            fsum = sum(f for _, f, _ in tower)
            ach = fsum / tsum / 1e12
            # HBM bytes per launch of the same kernels come from the rocprofv3 PMC passes of this very command
            # (FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 FETCH x2 correction; tools/pmc_summary.py) - PMC counters
            # cannot be read from inside the process, so a committed summary is quoted ONLY if it was measured on this very
            # build of the kernels (its recorded source hash == tools/pmc_summary.py:kernel_source_hash() now); else null.
            traffic, traffic_src = None, "null: no PMC summary of this kernel build under profiles/ (tools/profile_round.sh regenerates it)"
            if a.arch == ARCH and a.batch == BATCH and a.nodes == N_NODES:
                sys.path.insert(0, str(ROOT / "tools"))
                from pmc_summary import kernel_source_hash
                now = kernel_source_hash()
                for pmc in sorted((ROOT / "profiles").glob("r*_pmc_summary.json"), reverse=True):
                    d = json.load(open(pmc))
                    if d.get("kernel_source_hash") == now:
                        traffic = d["tower_gemm"]["hbm_bytes_per_launch"]
                        traffic_src = f"HBM bytes per launch (rocprofv3 PMC, profiles/{pmc.name}, kernel sources {now})"
                        break
            roof = {"kernel": "gemm_nt_256 / gemm_nt_128 (image-tower GEMMs: qkv, out, fc, proj, patch)" if a.arch.startswith("ViT") else "gemm_nt_256 / gemm_nt_128 (image-tower 1x1 GEMMs and implicit-GEMM 3x3 convolutions)", "bound": "mfma",
                    "achieved": round(ach, 1), "peak": PEAK_TFLOPS_BF16, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_BF16, 4),
                    "traffic": traffic, "traffic_unit": traffic_src,
                    "algorithmic_bytes_per_launch": round(sum(b for _, _, b in tower) / len(tower)),
                    "launches": len(tower), "avg_launch_us": round(tsum / len(tower) * 1e6, 1)}
            shapes = {}
            for (name, s_, e_, fl, by, tag) in recs:
                if name == "gemm_nt" and tag != "logits" and fl > 2e10:
                    shapes.setdefault(tag or "untagged", []).append((s_.elapsed_time(e_) * 1e-3, fl))
            roof["by_shape"] = {k: {"launches": len(v), "avg_us": round(sum(t for t, _ in v) / len(v) * 1e6, 1),
                                    "tflops": round(sum(f for _, f in v) / sum(t for t, _ in v) / 1e12, 1)} for k, v in sorted(shapes.items())}
        lg = [(s_.elapsed_time(e_) * 1e-3, fl, by) for (name, s_, e_, fl, by, tag) in recs if tag in ("logits", "logits_eval")]     # tagged, not guessed from sizes
        if lg and roof:
            tl = sum(t for t, _, _ in lg) / len(lg)
            roof["logits_gemm"] = {"us": round(tl * 1e6, 1), "tflops": round(lg[0][1] / tl / 1e12, 1),
                                   "frac_mfma": round(lg[0][1] / tl / 1e12 / PEAK_TFLOPS_BF16, 4),
                                   "gbps": round(lg[0][2] / tl / 1e9, 1), "frac_hbm": round(lg[0][2] / tl / 8e12, 4),
                                   "what": ("hgr_logits_eval: logits GEMM + top-20 / top-1 / level arg-max in its epilogue, both stages, no logits written "
                                            "(target of north_star: >= 0.40 of the MFMA peak)") if fused_eval else "hgr_gemm_nt writing fp32 logits (hgr_eval_rows runs behind it)"}

    # PCIe-inclusive rate (never `value`): the same step fed from pinned HOST memory with uint8 NHWC crops (what a
    # JPEG decoder hands over), H2D on a copy stream double-buffered against compute, normalisation fused into the
    # patch kernel (hgr_im2col_patches_u8).
    pcie = None
    if a.pcie and cfg["vision_patch_size"]:
        r = cfg["image_resolution"]
        g = torch.Generator().manual_seed(1234 + rank)
        host = [torch.randint(0, 256, (a.batch, r, r, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
        dbuf = [torch.empty((a.batch, r, r, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        cs = torch.cuda.Stream()
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        free = [torch.cuda.Event(), torch.cuda.Event()]
        for e_ in free:
            e_.record()

        def prefetch(i):
            with torch.cuda.stream(cs):
                cs.wait_event(free[i & 1])
                dbuf[i & 1].copy_(host[i & 1], non_blocking=True)
                ready[i & 1].record(cs)

        def ustep(i):
            prefetch(i + 1)
            torch.cuda.current_stream().wait_event(ready[i & 1])
            if fused_eval:
                ev.add_images(dbuf[i & 1], targets[i % len(targets)])
            else:
                logits = model(dbuf[i & 1], None, static_output=True)
                ev.add_batch(logits, targets[i % len(targets)], want_outputs=False)
            free[i & 1].record()

        prefetch(0)
        for i in range(min(3, a.warmup)):
            ustep(i)
        fence()
        base_i = min(3, a.warmup)
        t0 = time.perf_counter()
        for i in range(a.steps):
            ustep(base_i + i)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        pcie = {"value": round(a.batch * world * a.steps / el, 1), "unit": "images/sec", "ms_per_step": round(el / a.steps * 1e3, 3),
                "input": f"uint8 NHWC crops in pinned host memory ({a.batch * r * r * 3 / 1e6:.0f} MB/batch), H2D double-buffered on a copy stream"}

    cpu = parity = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(sd, model.zsl_weights.float().cpu(), arch=a.arch)
        # the metric's parity clause at full size: the HIP path on the oracle's batch (same weights, same class matrix)
        img_c, lg_c = cpu_baseline.last
        lg_g = model(img_c.to(dev), None).float().cpu()
        err = float((lg_g - lg_c).abs().max())
        te = model.test_index.cpu()
        sub_c, sub_g = lg_c[:, te], lg_g[:, te]
        top2 = sub_c.topk(2, dim=1)
        decidable = (top2.values[:, 0] - top2.values[:, 1]) > 2 * err          # outside the error band of the two paths
        same = sub_g.argmax(1) == top2.indices[:, 0]
        parity = {"images": int(img_c.shape[0]), "max_abs_logit_err": round(err, 6), "tolerance": 1e-3,
                  "hit1_equal": int(same.sum()), "hit1_decidable": int(decidable.sum()), "hit1_equal_decidable": int((same & decidable).sum()),
                  "note": "HIP logits (one batch of all these images) vs the CPU oracle's logits of every slice its timing leg ran, same workload; hit@1 over the test columns; "
                          "'decidable' = oracle top-1 margin > 2 x max logit error"}

    if rank == 0:
        line = {"metric": "images/sec over 21K-class hierarchy, batch 512; hit@1 parity vs reference",
                "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": a.image_dtype, "data": "synthetic",
                "config": {"workload": f"{a.arch} zero-shot eval step: encode_image + L2 + [{a.batch}x{cfg['embed_dim']}]x[{cfg['embed_dim']}x{a.nodes}] logits "
                                       f"+ top-20/top-1/level-argmax metrics (main.py:131-191; {'fused into the logits GEMM' if fused_eval else 'hgr_eval_rows on materialised logits'}), N={a.nodes} nodes, batch {a.batch}/GPU",
                           "global_batch": a.batch * world, "parallelism": f"dp{world}", "text_dtype": a.text_dtype,
                           "weights": "random-init (hash-seeded), no checkpoint offline"},
                "roofline": roof, "cpu_baseline": cpu, "parity": parity, "pcie_inclusive": pcie, "metrics_string": summary.strip()}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if group is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
