"""Multi-GPU sharding of the frame encoder.

Frames are independent units: rank r of W processes the 256-frame batches
{b : b mod W = r} (batch-interleaved, so temporal neighbours inside a batch stay on
one rank) with replicated weights and NO collective on the data path.  The one
exchange step is the all-gather of feature rows so that every rank holds the full
(N, F) sequence-feature matrix for the temporal / caption stage — it replaces the
reference's write-``.npy``-then-``np.load`` round trip (evaluate.py:316-321 ->
dataset.py:202-204).  ``torch.distributed`` backend "nccl" is RCCL over xGMI on
the GPU box; the same code runs on "gloo" in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def rank_batches(n_frames: int, batch: int, rank: int, world: int):
    """[(start, stop)] frame ranges owned by ``rank``."""
    nb = (n_frames + batch - 1) // batch
    return [(b * batch, min(n_frames, (b + 1) * batch)) for b in range(rank, nb, world)]


def local_rows(n_frames: int, batch: int, world: int) -> int:
    """Rows of the (padded) per-rank feature shard: equal on every rank."""
    nb = (n_frames + batch - 1) // batch
    return ((nb + world - 1) // world) * batch


def gather_feature_rows(shard: torch.Tensor, n_frames: int, batch: int, group=None) -> torch.Tensor:
    """All-gather per-rank shards (local_rows, F) and undo the batch interleave ->
    (n_frames, F) in global frame order, identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lr = local_rows(n_frames, batch, world)
    assert shard.shape[0] == lr, f"shard has {shard.shape[0]} rows, expected {lr}"
    if world == 1:
        return shard[:n_frames]
    f = shard.shape[1]
    gathered = torch.empty((world * lr, f), dtype=shard.dtype, device=shard.device)
    dist.all_gather_into_tensor(gathered, shard.contiguous(), group=group)
    # gathered[r, j, i] is frame (j*world + r)*batch + i  ->  order (j, r, i)
    out = gathered.view(world, lr // batch, batch, f).permute(1, 0, 2, 3).reshape(-1, f)
    return out[:n_frames]


def extract_features_sharded(encode_batch, n_frames: int, batch: int, feature_dim: int, device, rank=None,
                             world=None, group=None) -> torch.Tensor:
    """Run ``encode_batch(start, stop) -> (stop-start, F)`` over this rank's batches and
    all-gather the rows.  Returns the full (n_frames, F) matrix on every rank."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    shard = torch.zeros((local_rows(n_frames, batch, world), feature_dim), dtype=torch.float32, device=device)
    for j, (s, e) in enumerate(rank_batches(n_frames, batch, rank, world)):
        shard[j * batch: j * batch + (e - s)] = encode_batch(s, e)
    return gather_feature_rows(shard, n_frames, batch, group)
