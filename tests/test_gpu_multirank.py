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
