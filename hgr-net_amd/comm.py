"""RCCL collectives through libhgr's C ABI (include/hgr.h "Data-parallel collectives"): the lower-boundary alternative to
``torch.distributed`` for the three exchanges of the data-parallel path (hgr_net_amd.parallel): all-gather of class-embedding
row slices, all-reduce of the metric counters / the flat gradient buffer, broadcast of parameters.

The reference has no distributed code.  ``torch.distributed`` (backend "nccl" = RCCL) stays the default
transport of bench.py / main.py; this module is what a host that does not carry PyTorch's process groups binds.  In this
package HGR_COMM=native selects it for the blocking exchanges - the class-matrix all-gather (parallel.all_gather_rows), the
counter all-reduce (Evaluator.counters), FusedAdamW.allreduce and parallel.allreduce_grads; the overlapped gradient buckets
(FusedAdamW.allreduce_part) stay on torch.distributed's asynchronous work handles.  Bootstrap: rank 0 draws the 128-byte unique id and ships it over any side
channel - here an existing torch.distributed group (gloo is enough) or a file."""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import torch

from . import _lib

ID_BYTES = 128
F32, F64, F16, BF16, I32, I64, U8 = range(7)
SUM, MAX = 0, 1
_DT = {torch.float32: F32, torch.float64: F64, torch.float16: F16, torch.bfloat16: BF16, torch.int32: I32, torch.int64: I64, torch.uint8: U8}


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    _lib.call("hgr_comm_unique_id", buf)
    return buf.raw


def init(rank: int, world: int, uid: bytes) -> None:
    """Create the process's communicator on the CURRENT HIP device (call torch.cuda.set_device first)."""
    assert len(uid) == ID_BYTES
    torch.cuda.current_device()                       # make sure the HIP context of the chosen device exists
    _lib.call("hgr_comm_init", int(rank), int(world), C.create_string_buffer(uid, ID_BYTES))


def init_from_torch(group=None) -> None:
    """Bootstrap over an existing torch.distributed group: rank 0's id is broadcast as a Python object."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    init(rank, world, box[0])


def _run_nonce(nonce: Optional[str]) -> bytes:
    """16 bytes that identify THIS run: every rank derives them from what the launcher gave ALL of them - an explicit `nonce`, else
    HGR_COMM_NONCE (bench.py --gpus N and hgr_net_amd.main export a fresh one to their ranks), else MASTER_ADDR:MASTER_PORT +
    torchelastic's run id.  (Round 5 mixed in os.getppid() on one node; ranks started through a per-rank wrapper - numactl, a shell
    hop - then derived different nonces and timed out.  Nothing rank-local enters the nonce any more.)"""
    import hashlib
    if nonce is None:
        nonce = os.environ.get("HGR_COMM_NONCE") or ":".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"))
    return hashlib.sha256(nonce.encode()).digest()[:16]


def _nonce_is_per_run(nonce: Optional[str]) -> bool:
    """Whether the run nonce really differs from run to run: an explicit one (argument or HGR_COMM_NONCE) or a torchelastic run id
    other than the static default 'none'.  MASTER_ADDR:MASTER_PORT alone repeat across runs (29500) or are empty."""
    return bool(nonce or os.environ.get("HGR_COMM_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID", "none") not in ("", "none"))


def _process_start_time() -> float:
    """When THIS process was started (epoch seconds, /proc): the ranks of a run are started together, long before a slow first
    `import torch` finishes - so 'written after I was started' tells this run's id file from a crashed earlier run's without
    allowing for import time."""
    try:
        with open("/proc/self/stat") as f:
            fields = f.read().rsplit(")", 1)[1].split()            # fields[0] = field 3 (state); starttime = field 22
        with open("/proc/stat") as f:
            btime = next(int(l.split()[1]) for l in f if l.startswith("btime"))
        return btime + int(fields[19]) / os.sysconf("SC_CLK_TCK")
    except Exception:
        return time.time()


_T_START = _process_start_time()
# without a per-run nonce (static run id): an id file older than this process by more than HGR_COMM_STALE_S belongs to another run
# (ranks of one launch start within seconds of each other; round 5 measured from the module import and needed 300 s for slow imports)
STALE_SLACK_S = float(os.environ.get("HGR_COMM_STALE_S", "30"))


def init_from_file(path: str, rank: int, world: int, timeout_s: float = 120.0, nonce: Optional[str] = None) -> None:
    """Bootstrap over a shared file: rank 0 removes whatever an earlier run left at `path`, then publishes
    run-nonce + unique id atomically; the others wait for a file that carries THIS run's nonce (a stale or foreign id
    file is ignored instead of being fed to ncclCommInitRank, which would hang with no diagnostic).  The file is removed
    by rank 0's destroy()."""
    tagb = _run_nonce(nonce)
    per_run = _nonce_is_per_run(nonce)
    if not per_run and rank == 0:
        import warnings
        warnings.warn("hgr comm: no per-run nonce (pass nonce=, set HGR_COMM_NONCE, or launch with a torchelastic run id): a crashed run's id "
                      f"file at {path} is told apart by its age (written before this process started, by more than HGR_COMM_STALE_S = {STALE_SLACK_S:.0f} s = stale) "
                      "or by being replaced while a rank waits")
    if rank == 0:
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(tagb + unique_id())
        os.replace(tmp, path)
        _published.append(path)
    t0 = time.time()
    first_seen = None               # (inode, mtime) of the file this waiter found first: a file that CHANGES afterwards was written by this run
    while True:
        try:
            blob = open(path, "rb").read()
            st = os.stat(path)
            sig = (st.st_ino, st.st_mtime_ns)
            if first_seen is None:
                first_seen = sig
            fresh = per_run or rank == 0 or sig != first_seen or st.st_mtime >= _T_START - STALE_SLACK_S
        except FileNotFoundError:
            blob, fresh = b"", True
            first_seen = first_seen or (-1, -1)
        if len(blob) == 16 + ID_BYTES and blob[:16] == tagb and fresh:
            break
        if time.time() - t0 > timeout_s:
            what = "no unique id" if not blob else "only a stale unique id (another run's nonce, or older than this run)"
            raise _lib.HgrError(f"hgr comm: {what} at {path} after {timeout_s}s")
        time.sleep(0.05)
    init(rank, world, blob[16:])


_published: list = []


def destroy() -> None:
    _lib.call("hgr_comm_destroy")
    while _published:                                   # rank 0: the id file of this run is of no further use
        try:
            os.unlink(_published.pop())
        except OSError:
            pass


def world() -> int:
    return int(_lib.load().hgr_comm_world())


def rank() -> int:
    return int(_lib.load().hgr_comm_rank())


def active() -> bool:
    return world() > 0


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def allreduce(t: torch.Tensor, op: int = SUM) -> torch.Tensor:
    """In-place all-reduce of a contiguous device tensor on the current stream."""
    assert t.is_cuda and t.is_contiguous()
    _lib.call("hgr_allreduce", t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], op, _stream())
    return t


def allgather(local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = rank r's `local` (equal shapes on every rank); returns [world, *local.shape]."""
    assert local.is_cuda and local.is_contiguous()
    w = world()
    if out is None:
        out = torch.empty((w,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    assert out.is_contiguous() and out.numel() == w * local.numel() and out.dtype == local.dtype
    _lib.call("hgr_allgather", local.data_ptr(), out.data_ptr(), local.numel(), _DT[local.dtype], _stream())
    return out


def broadcast(t: torch.Tensor, root: int = 0) -> torch.Tensor:
    assert t.is_cuda and t.is_contiguous()
    _lib.call("hgr_broadcast", t.data_ptr(), t.numel(), _DT[t.dtype], int(root), _stream())
    return t
