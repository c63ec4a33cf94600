"""Multi-GPU sharding of the frame encoder (BASELINE config C4).

Frames are independent units: with one process per GPU, rank r of W owns the blocks
{k : k mod W = r} of ``block`` consecutive ``batch``-frame batches (block = 1: plain batch
interleave; temporal neighbours inside a batch stay on one rank), weights replicated,
NO collective on the data path.  The one exchange step is the all-gather of feature rows
so that every rank holds the full (N, F) sequence-feature matrix for the temporal /
caption stage — it replaces the reference's write-``.npy``-then-``np.load`` round trip
(evaluate.py:316-321 -> dataset.py:202-204; the reference splits each batch over its
``ctx`` list instead, evaluate.py:278-281,308-313).  The gather is CHUNKED and overlapped
with compute: as soon as a round of blocks is encoded its rows are all-gathered
(``async_op``) straight into their final place in the output — block k of the corpus lands
at rows [k·block·batch, (k+1)·block·batch), so no re-ordering pass follows — while the
next round is being encoded.  ``torch.distributed`` backend "nccl" is RCCL over xGMI on
the GPU box; the same code runs on "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
import socket

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------
# process bootstrap: one process per GPU
# ---------------------------------------------------------------------------------------
def init_distributed(backend: str | None = None):
    """(rank, world, device) of this process.  Under a launcher (RANK / WORLD_SIZE in the environment, set by
    ``torch.distributed.run`` or by ``launch`` below) the default process group is created — "nccl" (= RCCL) with the
    process bound to GPU LOCAL_RANK when a GPU is visible, "gloo" otherwise; without one it is (0, 1, current device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    has_gpu = torch.cuda.is_available()
    if has_gpu:
        if world > 1:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what this host driver supports
        backend = backend or ("nccl" if has_gpu else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, dev


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(local_rank, nprocs, port, fn, args):
    os.environ.update({"RANK": str(local_rank), "LOCAL_RANK": str(local_rank), "WORLD_SIZE": str(nprocs),
                       "LOCAL_WORLD_SIZE": str(nprocs), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    fn(*args)


def launch(fn, nprocs: int, args=()):
    """Run ``fn(*args)`` in ``nprocs`` fresh processes of this node (one per GPU), each with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) so that ``init_distributed`` forms the group:
    what ``python -m torch.distributed.run --nproc-per-node N`` does, without needing that launcher.  ``fn`` must be a
    module-level function (the workers are spawned, not forked: a HIP runtime does not survive a fork)."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(nprocs, free_port(), fn, tuple(args)), nprocs=nprocs, join=True)


def under_launcher() -> bool:
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


# ---------------------------------------------------------------------------------------
# the partition
# ---------------------------------------------------------------------------------------
def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def n_rounds(n_frames: int, batch: int, world: int, block: int = 1) -> int:
    """Rounds of (world x block) batches that cover the corpus."""
    per_round = world * block * batch
    return (n_frames + per_round - 1) // per_round


def rank_batches(n_frames: int, batch: int, rank: int, world: int, block: int = 1):
    """[(start, stop)] frame ranges owned by ``rank``, in the order it processes them."""
    out = []
    for c in range(n_rounds(n_frames, batch, world, block)):
        for j in range(block):
            s = ((c * world + rank) * block + j) * batch
            if s < n_frames:
                out.append((s, min(n_frames, s + batch)))
    return out


def local_rows(n_frames: int, batch: int, world: int, block: int = 1) -> int:
    """Rows of the (padded) per-rank feature shard: equal on every rank."""
    return n_rounds(n_frames, batch, world, block) * block * batch


def gather_feature_rows(shard: torch.Tensor, n_frames: int, batch: int, group=None, block: int = 1) -> torch.Tensor:
    """One-shot all-gather of complete per-rank shards (local_rows, F) -> (n_frames, F) in global frame order,
    identical on every rank."""
    world = _world(group)
    lr = local_rows(n_frames, batch, world, block)
    assert shard.shape[0] == lr, f"shard has {shard.shape[0]} rows, expected {lr}"
    if world == 1:
        return shard[:n_frames]
    f = shard.shape[1]
    gathered = torch.empty((world * lr, f), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, shard.contiguous(), group=group)
    # gathered[r, c, (j, i)] is frame ((c*world + r)*block + j)*batch + i  ->  order (c, r, (j, i))
    rows = block * batch
    out = gathered.view(world, lr // rows, rows, f).permute(1, 0, 2, 3).reshape(-1, f)
    return out[:n_frames]


def feature_comm(device, group=None):
    """The C-ABI communicator (tennis_amd.comm.Comm: RCCL behind ``tn_comm_*``) over the ranks of ``group`` when the
    process runs on a GPU, else None (the CPU tests exchange through torch.distributed / gloo)."""
    if torch.device(device).type != "cuda":
        return None
    from .comm import bring_up
    return bring_up(group, torch.device(device))


def extract_features_sharded(encode_batch, n_frames: int, batch: int, feature_dim: int, device, rank=None,
                             world=None, group=None, block: int = 1, stats: dict | None = None,
                             encode_into=None, join=None, comm=None) -> torch.Tensor:
    """Run ``encode_batch(start, stop) -> (stop-start, F)`` over this rank's batches and all-gather the rows round by
    round, each round's collective in flight while the next round is encoded.  Returns the full (n_frames, F) matrix
    on every rank.  ``stats`` (optional dict) receives ``rounds``, ``gather_bytes_per_rank`` and ``frames_local``.
    Pipelined encoders (``engine.DenseNet121Features.set_pipelined``): pass ``encode_into(start, stop, rows)`` - it
    writes its features into ``rows`` without waiting for them - and ``join()``, which orders everything encoded so far
    in front of what the current stream does next; it is called once per round, in front of the round's collective.
    ``comm`` (``feature_comm``): the collectives go through the library's own RCCL communicator (``tn_allgather_features``)
    instead of torch.distributed."""
    if rank is None:
        rank = _rank(group)
    if world is None:
        world = _world(group)
    rows = block * batch                                   # rows one rank contributes to a round
    rounds = n_rounds(n_frames, batch, world, block)
    out = torch.zeros((rounds * world * rows, feature_dim), dtype=torch.float32, device=device)
    # world 1: the "shard" is the output itself; otherwise a per-rank staging shard that the collectives read
    shard = out if world == 1 else torch.zeros((rounds * rows, feature_dim), dtype=torch.float32, device=device)
    pending, done = [], 0
    for c in range(rounds):
        for j in range(block):
            s = ((c * world + rank) * block + j) * batch
            if s >= n_frames:
                break
            e = min(n_frames, s + batch)
            r0 = c * rows + j * batch
            if encode_into is not None:
                encode_into(s, e, shard[r0:r0 + (e - s)])
            else:
                shard[r0:r0 + (e - s)] = encode_batch(s, e)
            done += e - s
        if join is not None:
            join()       # every round (a few stream waits): bounds what a pipelined encoder keeps referenced, also with one rank
        if world > 1:
            # rows [c*world*rows, (c+1)*world*rows) of the output = rank-major concatenation of this round's shard chunks
            if comm is not None:
                pending.append(comm.allgather_features(shard[c * rows:(c + 1) * rows], out[c * world * rows:(c + 1) * world * rows]))
            else:
                pending.append(dist.all_gather_into_tensor(out[c * world * rows:(c + 1) * world * rows],
                                                           shard[c * rows:(c + 1) * rows], group=group, async_op=True))
            if len(pending) > 4:                           # bound the collectives in flight
                pending.pop(0).wait()
    for w in pending:
        w.wait()
    if stats is not None:
        stats.update(rounds=rounds, frames_local=done, gather_bytes_per_rank=(rounds * rows * feature_dim * 4 if world > 1 else 0))
    return out[:n_frames]
