"""RCCL communicator of the C-ABI (include/tennis_hip.h ``tn_comm_*``; csrc/comm.hip): the path's one exchange step -
the all-gather of feature rows over xGMI (BASELINE config C4) - and the all-reduce of the trainers' flat gradient
buffers / confusion counts, one process per GPU.

It replaces the reference's exchange medium: per-frame ``.npy`` files written by ``save_features``
(evaluate.py:306-321) and read back by the temporal stage (dataset.py:202-204), and the kvstore all-reduce behind
``gluon.Trainer.step`` over a ctx list (train.py:410-424).

The collectives run on a stream of their own (ordered behind whatever the caller's current stream has queued when a
call is made), so a round's all-gather overlaps the encoding of the next round; ``Handle.wait()`` orders the caller's
stream behind the collective.  torch.distributed is used for one thing only: handing rank 0's 128-byte unique id to the
other ranks (any out-of-band channel would do - a maintainer binding the ABI from another host language uses a file or
their launcher's store)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class Handle:
    """Completion of one collective on the communicator's stream."""

    def __init__(self, event: torch.cuda.Event):
        self._event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self._event)


class Comm:
    def __init__(self, rank: int = 0, world: int = 1, unique_id: bytes | None = None, device: int | None = None,
                 force_rccl: bool = False):
        self.rank, self.world = int(rank), int(world)
        dev = _lib.default_device() if device is None else int(device)
        self.stream = torch.cuda.Stream(device=dev)
        self.ctx = _lib.Context(dev, stream=self.stream)
        self.lib = self.ctx.lib
        h = _lib._P()
        idbuf = (C.c_char * 128).from_buffer_copy(unique_id) if unique_id is not None else None
        _lib.check(self.lib.tn_comm_create(self.ctx.handle, self.rank, self.world, idbuf, 1 if force_rccl else 0, C.byref(h)),
                   "tn_comm_create")
        self.handle = h

    # -- bootstrap ------------------------------------------------------------------------------------------------
    @staticmethod
    def new_unique_id() -> bytes:
        buf = (C.c_char * 128)()
        _lib.check(_lib.load().tn_comm_unique_id(buf), "tn_comm_unique_id")
        return bytes(buf)

    @classmethod
    def from_process_group(cls, group=None, device: int | None = None):
        """A communicator over the ranks of a torch.distributed group (default: the world); the unique id travels through
        the group's object broadcast."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return cls(0, 1, None, device)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0], device)

    # -- collectives ----------------------------------------------------------------------------------------------
    def _enter(self, *tensors):
        for t in tensors:
            assert t.is_cuda and t.is_contiguous() and t.device.index == self.ctx.device, "comm: contiguous tensors on the communicator's GPU"
            t.record_stream(self.stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.ctx.device))

    def _leave(self) -> Handle:
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return Handle(ev)

    def allgather_features(self, shard: torch.Tensor, out: torch.Tensor) -> Handle:
        """out (world * rows, F) <- rank-major concatenation of every rank's shard (rows, F), fp32."""
        rows, f = shard.shape
        assert shard.dtype == torch.float32 and out.dtype == torch.float32 and tuple(out.shape) == (self.world * rows, f)
        self._enter(shard, out)
        _lib.check(self.lib.tn_allgather_features(self.handle, _lib.ptr(shard), rows, f, _lib.ptr(out)), "tn_allgather_features")
        return self._leave()

    def allreduce_(self, buf: torch.Tensor, average: bool = False) -> Handle:
        """in-place sum (mean) over the ranks: fp32 or int64."""
        self._enter(buf)
        if buf.dtype == torch.float32:
            _lib.check(self.lib.tn_allreduce_f32(self.handle, _lib.ptr(buf), buf.numel(), 1 if average else 0), "tn_allreduce_f32")
        elif buf.dtype == torch.int64 and not average:
            _lib.check(self.lib.tn_allreduce_i64(self.handle, _lib.ptr(buf), buf.numel()), "tn_allreduce_i64")
        else:
            raise TypeError(f"comm.allreduce_: unsupported dtype {buf.dtype}")
        return self._leave()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tn_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
