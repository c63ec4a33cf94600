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
        # rank 0 ALWAYS takes part in the broadcast: if it cannot make an id (dlopen of librccl, ncclGetUniqueId) it sends the
        # error instead, and every rank raises behind the broadcast - so that all ranks leave this function together and
        # reach bring_up()'s agreement step (a rank 0 that raised in front of the broadcast would be inside that step's
        # all-reduce while the others still sit in the broadcast: mismatched collectives, a hang instead of a fall-back)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.new_unique_id()
            except Exception as e:
                box[0] = ("error", repr(e))
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], tuple):
            raise RuntimeError(f"rank 0 could not create a communicator id: {box[0][1]}")
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

    def describe(self) -> dict:
        """what the communicator reports about itself (RCCL's own answers when RCCL is behind the handle: tn_comm_world = ncclCommCount ...)"""
        return {"transport": self.transport, "rank": int(self.lib.tn_comm_rank(self.handle)), "world": int(self.lib.tn_comm_world(self.handle)),
                "device": int(self.lib.tn_comm_device(self.handle)), "rccl": bool(self.lib.tn_comm_uses_rccl(self.handle))}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tn_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupComm:
    """The same two collectives through torch.distributed (backend nccl = RCCL): what ``feature_comm`` hands out when the
    library's own communicator cannot be brought up on EVERY rank of a multi-rank job (agreed by an all-reduce, so no rank
    is left waiting inside a collective the others never enter).  It is the same RCCL library and the same collectives, only
    issued through torch's communicator; the reason is printed once, by rank 0, and ``bench.py`` names the transport it
    used in ``config.exchange``."""
    transport = "torch.distributed (RCCL)"

    class _Work:
        def __init__(self, work):
            self._work = work

        def wait(self):
            self._work.wait()

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather_features(self, shard: torch.Tensor, out: torch.Tensor):
        return self._Work(self._dist.all_gather_into_tensor(out, shard, group=self.group, async_op=True))

    def allreduce_(self, buf: torch.Tensor, average: bool = False):
        w = self._dist.all_reduce(buf, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)
        if average:
            w.wait()
            buf /= self.world
        return self._Work(w)

    def describe(self) -> dict:
        import torch
        return {"transport": self.transport, "rank": self.rank, "world": self.world,
                "device": torch.cuda.current_device() if torch.cuda.is_available() else -1, "rccl": self._dist.get_backend(self.group) == "nccl"}

    def close(self):
        pass


Comm.transport = "tn_allgather_features (librccl behind the C ABI)"


def bring_up(group=None, device=None, _make=None):
    """``Comm.from_process_group`` with a handshake: every rank reports whether its communicator came up AND a probe
    all-gather returned every rank's number; only if all did is the library communicator used, otherwise all ranks fall
    back to ``GroupComm`` together.  (``_make``: the factory, replaced by the gloo test of this handshake.)"""
    import sys
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    index = device.index if isinstance(device, torch.device) else device
    make = _make or (lambda: Comm.from_process_group(group, index))
    comm, err = None, ""
    try:
        comm = make()
    except Exception as e:          # dlopen of librccl, ncclCommInitRank ...
        err = repr(e)
    if not multi:
        if comm is None:
            raise RuntimeError(err)
        return comm
    dev = device if isinstance(device, torch.device) else torch.device("cuda", _lib.default_device() if device is None else int(device))

    def agreed(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    # 1. did every rank get a communicator?  (a rank without one must not leave the others waiting inside the probe)
    ok = agreed(comm is not None)
    # 2. does a collective on it deliver every rank's rows?
    if ok:
        try:
            mine = torch.full((1, 4), float(comm.rank), dtype=torch.float32, device=dev)
            allr = torch.full((comm.world, 4), -1.0, dtype=torch.float32, device=dev)
            comm.allgather_features(mine, allr).wait()
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            good = bool(torch.equal(allr[:, 0].cpu(), torch.arange(comm.world, dtype=torch.float32)))
            err = err or ("" if good else "probe all-gather returned wrong rows")
        except Exception as e:
            good, err = False, repr(e)
        ok = agreed(good)
    if ok:
        return comm
    if comm is not None:
        comm.close()
    if dist.get_rank(group) == 0 or err:
        print(f"tennis_amd.comm: library communicator not usable on every rank ({err or 'another rank failed'}); "
              "the exchange step goes through torch.distributed's RCCL communicator instead", file=sys.stderr, flush=True)
    return GroupComm(group)
