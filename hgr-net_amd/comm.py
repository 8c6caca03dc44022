"""RCCL collectives through libhgr's C ABI (include/hgr.h "Data-parallel collectives"): the lower-boundary alternative to
``torch.distributed`` for the three exchanges of the data-parallel path (hgr_net_amd.parallel): all-gather of class-embedding
row slices, all-reduce of the metric counters / the flat gradient buffer, broadcast of parameters.

The reference has no distributed code.  ``torch.distributed`` (backend "nccl" = RCCL) stays the default
transport of bench.py / main.py; this module is what a host that does not carry PyTorch's process groups binds, and it is
selected in this package with HGR_COMM=native.  Bootstrap: rank 0 draws the 128-byte unique id and ships it over any side
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


def init_from_file(path: str, rank: int, world: int, timeout_s: float = 120.0) -> None:
    """Bootstrap over a shared file: rank 0 writes the id (atomically), the others wait for it."""
    if rank == 0:
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(unique_id())
        os.replace(tmp, path)
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout_s:
            raise _lib.HgrError(f"hgr comm: no unique id at {path} after {timeout_s}s")
        time.sleep(0.05)
    init(rank, world, open(path, "rb").read())


def destroy() -> None:
    _lib.call("hgr_comm_destroy")


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
