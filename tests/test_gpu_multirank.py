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
    assert p.returncode == 0, p.stderr[-2000:]
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
    assert set(d1) == set(d2)


def _dp(mode, world, out):
    """Run tests/workers/dp_worker.py with `world` ranks (children of torch.distributed.run for world > 1)."""
    worker = str(ROOT / "tests" / "workers" / "dp_worker.py")
    env = dict(os.environ, HGR_TEST_ONE_GPU="1")
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


def test_two_rank_om_step_equals_one_rank(tmp_path):
    """SURVEY H7 (model/clip_tree.py:222-281): ONE single-class batch sharded over 2 ranks with identical sampling seeds,
    mean CE over equal shards, gradient all-reduce(sum) x 1/world == the 1-rank full-batch step.
    Image tower: every row's backward is bit-identical up to the exact factor 2 of the shard's 1/b, so the averaged
    gradient differs from the full-batch one by fp32 summation order only (<= 1e-4 relative L2).  Text tower: each rank
    back-propagates ITS partial feature gradient and the backward chain rounds activations' gradients to bf16 (nonlinear),
    so the bound there is bf16 rounding (2^-9 per element, random; measured 1.8e-2 worst): <= 4e-2 relative L2, cosine >= 0.999."""
    import torch
    one = _dp("train", 1, tmp_path / "t1.pt")
    two = _dp("train", 2, tmp_path / "t2.pt")
    assert one["contra"] == two["contra"]                      # same negatives on every rank
    assert abs(one["loss"] - two["loss"]) <= 1e-5 * abs(one["loss"]), (one["loss"], two["loss"])
    worst_img, worst_txt = 0.0, 0.0
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
            cos = float(torch.dot(g1.flatten(), g2.flatten()) / (g1.norm() * g2.norm()))
            assert rel <= 4e-2 and cos >= 0.999, (k, rel, cos)
    print(f"\n[2-rank vs 1-rank OM step] worst relative L2 difference: image tower {worst_img:.2e}, text tower {worst_txt:.2e}")


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
