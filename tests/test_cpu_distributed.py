"""-m "not gpu": the N>1 path (frame sharding + feature all-gather) on gloo, world_size 2."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tennis_amd import sharding


def _worker(rank, world, port, n_frames, batch, fdim, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def encode(s, e):  # stand-in encoder: a recognisable function of the global frame index
        idx = torch.arange(s, e, dtype=torch.float32)
        return idx[:, None] * 10 + torch.arange(fdim, dtype=torch.float32)[None, :]
    full = sharding.extract_features_sharded(encode, n_frames, batch, fdim, "cpu")
    ret[rank] = full.numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_extract_allgather_world2():
    n_frames, batch, fdim, world = 37, 8, 5, 2      # ragged: 5 batches, last one short, uneven per rank
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29531, n_frames, batch, fdim, ret), nprocs=world, join=True)
        want = np.arange(n_frames, dtype=np.float32)[:, None] * 10 + np.arange(fdim, dtype=np.float32)[None, :]
        for r in range(world):
            assert np.array_equal(ret[r], want)


def test_rank_batches_partition():
    for n, b, w in [(786455, 256, 8), (37, 8, 2), (5, 8, 4), (256, 256, 1)]:
        seen = []
        for r in range(w):
            seen += sharding.rank_batches(n, b, r, w)
        seen.sort()
        assert seen[0][0] == 0 and seen[-1][1] == n
        assert all(a[1] == c[0] for a, c in zip(seen, seen[1:]))
        assert sharding.local_rows(n, b, w) * w >= n
