"""GPU: the multi-rank control flow of bench.py (sharded text encode + all-gather, batches dealt over ranks, counter
all-reduce, max-over-ranks timing) with two ranks sharing the one GPU of the test box (gloo backend: RCCL refuses two
ranks on one device; the driver's real N > 1 runs use RCCL).  Run as child processes under torch.distributed.run."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(n, extra):
    env = dict(os.environ, HGR_TEST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(n)] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and any(w in p.stderr for w in ("Address already in use", "RendezvousConnectionError", "DistNetworkError")):
        pytest.skip("could not set up a local rendezvous on this box")
    why = [l for l in p.stderr.splitlines() if "dp_check FAILED" in l or "Error" in l][:6]
    assert p.returncode == 0, "\n".join(why) + "\n" + p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]                      # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def test_two_ranks_complete_and_report_the_job():
    """1 rank and 2 ranks (torch.distributed.run, as the driver launches it): both finish, exactly one JSON line from
    rank 0, same keys, n_gpus / global batch / parallelism describe the whole job."""
    common = ["--steps", "8", "--warmup", "2", "--nodes", "3000", "--batch", "32", "--arch", "small-vit", "--no-cpu-baseline", "--no-pcie"]
    one = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    d2 = _bench(2, common)
    assert d2["n_gpus"] == 2 and d2["config"]["global_batch"] == 64 and d2["config"]["parallelism"] == "dp2" and d2["value"] > 0
    assert d1["n_gpus"] == 1 and d1["scaling"] == d2["scaling"] == "weak"
    assert set(d1) | {"dp_check"} == set(d2)
    # the N > 1 line verifies itself (bench.py:dp_check_eval): class matrix identical on all ranks and equal to each rank's own
    # encoding of a row sample, all-reduced counters == sum of the per-rank counters, both ranks seen
    c = d2["dp_check"]
    assert c["ok"] and c["ranks_seen"] == 2 and c["zsl_checksum_equal"] and c["counters_allreduce_equals_sum_of_ranks"], c
    assert c["own_encode_max_abs_diff"] <= 1e-6 and sum(c["num_sample_per_rank"]) == 2 * (8 + 2) * 32, c      # warm-up batches are counted too
    for d in (d1, d2):                                          # planted signal: the counters of main.py:139-191 actually move
        assert d["planted_signal"]["planted_rows"] > 0 and any(ch in "123456789" for ch in d["metrics_string"]), d["metrics_string"]


def test_two_rank_training_bench_verifies_itself():
    """`bench.py --mode train` on two ranks: the JSON line carries dp_check (gradient buffer and weights identical on both ranks
    after the overlapped all-reduce + fused AdamW, same negatives), and the run exits 0 only when it holds."""
    d = _bench(2, ["--mode", "train", "--arch", "small-vit", "--nodes", "3000", "--batch", "8", "--steps", "2", "--warmup", "1"])
    c = d["dp_check"]
    assert c["ok"] and c["ranks_seen"] == 2 and c["grad_buffer_checksum_equal"] and c["weights_checksum_equal"] and c["negatives_equal"], c
    assert len(c["loss_per_rank"]) == 2 and d["n_gpus"] == 2


def _dp(mode, world, out, **extra_env):
    """Run tests/workers/dp_worker.py with `world` ranks (children of torch.distributed.run for world > 1)."""
    worker = str(ROOT / "tests" / "workers" / "dp_worker.py")
    env = dict(os.environ, HGR_TEST_ONE_GPU="1", **extra_env)
    if world == 1:
        cmd = [sys.executable, worker, mode, str(out)]
        env.pop("WORLD_SIZE", None)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), worker, mode, str(out)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if p.returncode != 0 and any(w in p.stderr for w in ("Address already in use", "RendezvousConnectionError", "DistNetworkError")):
        pytest.skip("could not set up a local rendezvous on this box")
    assert p.returncode == 0, p.stderr[-3000:]
    import torch
    return torch.load(out, weights_only=False)


def test_two_rank_evaluation_equals_one_rank(tmp_path):
    """the metric counters of main.test (main.py:121-128,139-191) from 2 ranks - sharded text encode +
    all-gather, batches dealt round-robin, ONE counter all-reduce - equal the 1-rank counters: integer counters exactly,
    the two fraction accumulators (sums of edge/(L-1), point/L per batch: the all-reduce adds them in another order)
    to 1e-12, the printed metric string identical, and the all-gathered class matrix bit-identical."""
    import torch
    one = _dp("eval", 1, tmp_path / "e1.pt")
    two = _dp("eval", 2, tmp_path / "e2.pt")
    assert torch.equal(one["zsl"], two["zsl"])
    for k, v in one["counters"].items():
        if k in ("path_all", "point_all"):
            assert abs(two["counters"][k] - v) <= 1e-12 * max(1.0, abs(v)), (k, v, two["counters"][k])
        else:
            assert two["counters"][k] == v, (k, v, two["counters"][k])
    assert one["counters"]["num_sample"] == 7 * 16
    assert one["summary"] == two["summary"]


def test_two_rank_om_step_with_differently_seeded_ranks(tmp_path):
    """Round-4 advisor finding: hgr_net_amd.main seeded Python's `random` on no rank, so the ranks of a data-parallel OM step drew
    different negatives and the prompt-parallel text tower raised at the first step.  Ranks seeded 5 and 6 here: the step completes,
    warns once, and every rank has worked on rank 0's draw (main.run now also seeds all ranks from one broadcast value)."""
    import math
    two = _dp("train", 2, tmp_path / "u2.pt", HGR_TEST_UNSEEDED="1", HGR_DP_EXACT_HEAD="0")
    assert len(two["contra_per_rank"]) == 2 and two["contra_per_rank"][0] == two["contra_per_rank"][1] == two["contra"]
    assert math.isfinite(two["loss"]) and two["loss"] > 0.0
    assert all(bool(g.isfinite().all()) for g in two["grads"].values())


@pytest.mark.parametrize("exact_head", [False, True])
def test_two_rank_om_step_equals_one_rank(tmp_path, exact_head):
    """SURVEY H7 (model/clip_tree.py:222-281): ONE single-class batch sharded over 2 ranks with identical sampling seeds,
    mean CE over equal shards, gradient all-reduce(sum) x 1/world == the 1-rank full-batch step.
    Image tower: every row's backward is bit-identical up to the exact factor 2 of the shard's 1/b, so the averaged
    gradient differs from the full-batch one by fp32 summation order only (<= 1e-4 relative L2).
    Text tower (round 4): the step's distinct prompts are sharded over the ranks and every rank back-propagates the WHOLE batch's
    feature gradient of its own prompts (round 3 back-propagated each rank's partial gradient: 1.8e-2).
      * default (feature gradient all-reduced): the sum over ranks is associated differently from the one-rank matmul, and a bf16
        backward chain amplifies any input difference - even 1e-7 - to its own rounding noise within a few layers (a flipped
        rounding is a 2^-9 error that flips more roundings downstream): measured 3 - 4e-3, bound 8e-3;
      * HGR_DP_EXACT_HEAD=1 (image features all-gathered, the loss head of the whole batch on every rank): the text backward starts
        from the one-rank step's bits x 2, and the text tower meets the image tower's bound, 1e-4.
    Either way every rank encodes 1 / world of the prompts."""
    import torch
    extra = {"HGR_DP_EXACT_HEAD": "1"} if exact_head else {"HGR_DP_EXACT_HEAD": "0"}
    one = _dp("train", 1, tmp_path / "t1.pt", **extra)
    two = _dp("train", 2, tmp_path / "t2.pt", **extra)
    assert one["contra"] == two["contra"]                      # same negatives on every rank
    assert abs(one["loss"] - two["loss"]) <= 1e-5 * abs(one["loss"]), (one["loss"], two["loss"])
    worst_img, worst_txt = 0.0, 0.0
    report = {k: float((g1 - two["grads"][k]).norm() / max(float(g1.norm()), 1e-30)) for k, g1 in one["grads"].items() if float(g1.norm()) >= 1e-9}
    print("\n[2-rank vs 1-rank OM step] relative L2 per parameter (worst 12): " + ", ".join(f"{k} {v:.1e}" for k, v in sorted(report.items(), key=lambda kv: -kv[1])[:12]))
    for k, g1 in one["grads"].items():
        g2 = two["grads"][k]
        n1 = float(g1.norm())
        if n1 < 1e-9:
            assert float(g2.norm()) < 1e-6, k
            continue
        rel = float((g1 - g2).norm()) / n1
        if k.startswith("visual."):
            worst_img = max(worst_img, rel)
            assert rel <= 1e-4, (k, rel)
        elif k == "logit_scale":
            assert abs(float(g1) - float(g2)) <= 1e-4 * (1.0 + abs(float(g1))), (float(g1), float(g2))
        else:
            worst_txt = max(worst_txt, rel)
            assert rel <= (1e-4 if exact_head else 8e-3), (k, rel)
    print(f"\n[2-rank vs 1-rank OM step] worst relative L2 difference: image tower {worst_img:.2e}, text tower {worst_txt:.2e}")
    # prompt-parallel: rank 0 of the 2-rank run encoded half of the distinct prompts (ceil), and launched about half of the step's
    # GEMM FLOPs (half of the images, half of the prompts; the [B, <= 257] loss heads are fp32 matmuls outside these records)
    assert one["uniq"] == two["uniq"] and one["text_rows"] == one["uniq"] and two["text_rows"] == -(-two["uniq"] // 2), (one["text_rows"], two["text_rows"], one["uniq"])
    assert 0.45 * one["gemm_flops"] <= two["gemm_flops"] <= 0.56 * one["gemm_flops"], (one["gemm_flops"], two["gemm_flops"])


def test_two_rank_coop_context_gradient_is_identical_on_both_ranks_and_equals_one_rank(tmp_path):
    """BASELINE configs[4] (CoOp learnable context, model/CoOp.py:58-113) under the prompt-parallel text tower: d loss / d ctx sums
    over ALL prompts of the step, each rank back-propagates its own half, the bucketed all-reduce adds the halves - after it both
    ranks hold the same bits, equal to the one-rank full-batch gradient up to fp32 summation order."""
    import torch
    one = _dp("trainctx", 1, tmp_path / "c1.pt", HGR_DP_EXACT_HEAD="1")
    two = _dp("trainctx", 2, tmp_path / "c2.pt", HGR_DP_EXACT_HEAD="1")
    assert one["contra"] == two["contra"]
    assert abs(one["loss"] - two["loss"]) <= 1e-5 * abs(one["loss"]), (one["loss"], two["loss"])
    a, b = two["ctx_grad_per_rank"]
    assert torch.equal(a, b)
    ref = one["ctx_grad_per_rank"][0]
    assert float(ref.norm()) > 0
    rel = float((a - ref).norm() / ref.norm())
    assert rel <= 1e-4, rel
    for k, g1 in one["grads"].items():
        if not k.startswith("visual.") and k != "logit_scale" and float(g1.norm()) > 1e-9:
            assert float((g1 - two["grads"][k]).norm() / g1.norm()) <= 1e-4, k


def test_many_rank_bench_control_flow_on_one_gpu():
    """`bench.py --gpus 8` (evaluation) with eight ranks sharing the test box's one GPU (gloo; toy size; N % 8 != 0): the 8-way control
    flow - shard bounds of the prompt shards with a short last shard, batches dealt to 8 ranks, counter all-reduce, dp_check, exactly
    one JSON line - runs before the driver's real 8-GPU node ever does.  The test boxes allow at most 6 processes on the card at once:
    this pytest process and the torch.distributed.run agent are two of them (a 5-rank run was killed by the box's process guard), so
    the default is 4 ranks on class / prompt counts that 4 does not divide; HGR_TEST_MAX_RANKS=8 runs the full eight where nothing
    limits it."""
    n = int(os.environ.get("HGR_TEST_MAX_RANKS", "4"))
    d = _bench(n, ["--steps", "3", "--warmup", "1", "--nodes", "1003", "--batch", "8", "--arch", "small-vit", "--no-cpu-baseline", "--no-pcie"])
    c = d["dp_check"]
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == 8 * n
    assert c["ok"] and c["ranks_seen"] == n and c["zsl_checksum_equal"] and c["counters_allreduce_equals_sum_of_ranks"], c
    assert len(c["num_sample_per_rank"]) == n and sum(c["num_sample_per_rank"]) == n * (3 + 1) * 8, c
    # and one sharded training step per rank count: the prompt shards of `n` ranks over a prompt list that does not divide evenly
    t = _bench(n, ["--mode", "train", "--arch", "small-vit", "--nodes", "1003", "--batch", "2", "--steps", "1", "--warmup", "1"])
    assert t["dp_check"]["ok"] and t["dp_check"]["negatives_equal"] and t["n_gpus"] == n, t["dp_check"]


def test_two_rank_adaptive_layer_weights_stay_identical(tmp_path):
    """Round-2 advisor finding: with --weights adaptive the all-reduce of layer_weight.grad (a leaf outside the flat gradient
    buffer) ran on the first step only, so from step 2 (overlapped hooks installed) every rank stepped its own copy.
    hgr_net_amd.main.train for three steps on two ranks: layer_weight moved, is bit-identical on both ranks, and tracks the
    1-rank full-batch run (same global batch, same seeds) to the text-tower tolerance."""
    import torch
    two = _dp("adaptive", 2, tmp_path / "a2.pt")
    one = _dp("adaptive", 1, tmp_path / "a1.pt")
    a, b = two["layer_weight"]
    assert torch.equal(a, b), (a, b)
    assert float((a - two["initial"]).abs().max()) > 0, "layer_weight never moved: the test does not exercise optimizer2"
    ref = one["layer_weight"][0]
    moved = float((ref - one["initial"]).abs().max())
    assert float((a - ref).abs().max()) <= 5e-2 * moved + 1e-7, (a, ref, moved)


def _dp_native(world, out):
    worker = str(ROOT / "tests" / "workers" / "dp_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("HGR_TEST_ONE_GPU", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MASTER_PORT"] = str(_free_port())
    if world == 1:
        cmd = [sys.executable, worker, "native", str(out)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", env["MASTER_PORT"], worker, "native", str(out)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import torch
    return torch.load(out, weights_only=False)


def test_native_rccl_collectives_match_torch_distributed(tmp_path):
    """Round-2 advisor finding: hgr_allreduce / hgr_allgather / hgr_broadcast (csrc/hgr_comm.hip over dlopen'ed librccl) had only
    argument-validation coverage.  They run here against torch.distributed on the same data: on one rank always (dlopen, communicator
    init, datatype / op enum mapping, stream ordering), and on two ranks when the box has two devices (one rank per device)."""
    import torch
    res = _dp_native(1, tmp_path / "n1.pt")
    assert res and all(res.values()), res
    if torch.cuda.device_count() >= 2:
        res2 = _dp_native(2, tmp_path / "n2.pt")
        assert res2 and all(res2.values()), res2


def test_native_comm_carries_the_bench_exchanges():
    """HGR_COMM=native: the class-matrix all-gather and the counter all-reduce of bench.py go through libhgr's communicator
    (world 1 under HGR_FORCE_DIST=1: the only RCCL geometry a one-GPU box admits); same line as the torch.distributed transport."""
    common = ["--steps", "4", "--warmup", "1", "--nodes", "2000", "--batch", "16", "--arch", "small-vit", "--no-cpu-baseline", "--no-pcie"]
    outs = []
    for native in (False, True):
        env = {k: v for k, v in os.environ.items() if k not in ("HGR_TEST_ONE_GPU", "HGR_COMM")}
        env.update(HGR_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        if native:
            env["HGR_COMM"] = "native"
        p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0]))
    assert outs[0]["metrics_string"] == outs[1]["metrics_string"] and outs[1]["ranks_seen"] == 1


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no launcher, WORLD_SIZE unset): the parent spawns the two ranks as children of
    torch.distributed.run BEFORE any GPU call, relays rank 0's single JSON line and returns the children's status."""
    env = dict(os.environ, HGR_TEST_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--nodes", "2000", "--batch", "16",
                        "--arch", "small-vit", "--no-cpu-baseline", "--no-pcie"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and any(w in p.stderr for w in ("Address already in use", "RendezvousConnectionError", "DistNetworkError")):
        pytest.skip("could not set up a local rendezvous on this box")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["parallelism"] == "dp2"
