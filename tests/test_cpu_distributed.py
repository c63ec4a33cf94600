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


class _FakeLibComm:
    """Stands in for tennis_amd.comm.Comm on gloo (the library communicator needs GPUs): same surface, optionally broken."""
    transport = "fake library communicator"

    def __init__(self, rank, world, wrong_rows=False):
        self.rank, self.world, self.wrong, self.closed = rank, world, wrong_rows, False

    class _H:
        def wait(self):
            pass

    def allgather_features(self, shard, out):
        dist.all_gather_into_tensor(out, shard)
        if self.wrong:
            out[0] = 99.0
        return self._H()

    def close(self):
        self.closed = True


def _bringup_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tennis_amd import comm as cm

    def make():
        if mode == "rank1_raises" and rank == 1:
            raise RuntimeError("ncclCommInitRank failed (simulated)")
        return _FakeLibComm(rank, world, wrong_rows=(mode == "rank0_wrong_rows" and rank == 0))
    if mode == "rank0_id_fails":
        # the real factory (Comm.from_process_group), with the two calls that need librccl / a GPU replaced: rank 0 cannot make
        # a unique id (dlopen of librccl failing there).  It must still enter the id broadcast, or rank 1 waits in it forever
        # while rank 0 sits in the agreement all-reduce.
        def no_id():
            raise OSError("librccl.so: cannot open shared object file (simulated)")
        cm.Comm.new_unique_id = staticmethod(no_id)
        cm.Comm.__init__ = lambda self, *a, **k: (_ for _ in ()).throw(AssertionError("no communicator may be built without an id"))
        make = None
    c = cm.bring_up(None, torch.device("cpu"), _make=make)
    # whatever was agreed on must work on every rank
    mine = torch.full((2, 3), float(rank))
    allr = torch.empty((2 * world, 3))
    c.allgather_features(mine, allr).wait()
    ret[rank] = (type(c).__name__, allr[:, 0].tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_comm_bring_up_handshake_world2():
    """comm.bring_up: the library communicator is used only if it came up AND passed a probe all-gather on EVERY rank; if one
    rank fails to create it, or one rank's probe returns wrong rows, all ranks switch to the torch.distributed transport
    together (no rank is left inside a collective the others never enter) - also when rank 0 cannot even make the unique id
    (it used to raise in front of the id broadcast)."""
    for port, mode, want in ((29541, "ok", "_FakeLibComm"), (29542, "rank1_raises", "GroupComm"), (29543, "rank0_wrong_rows", "GroupComm"),
                             (29544, "rank0_id_fails", "GroupComm")):
        with mp.Manager() as mgr:
            ret = mgr.dict()
            mp.spawn(_bringup_worker, args=(2, port, mode, ret), nprocs=2, join=True)
            for r in range(2):
                assert ret[r][0] == want, (mode, ret[r])
                assert ret[r][1] == [0.0, 0.0, 1.0, 1.0] or mode == "ok"


def test_rank_batches_partition():
    for n, b, w, blk in [(786455, 256, 8, 1), (786455, 256, 8, 4), (37, 8, 2, 1), (37, 8, 2, 3), (5, 8, 4, 2), (256, 256, 1, 1)]:
        seen = []
        for r in range(w):
            mine = sharding.rank_batches(n, b, r, w, blk)
            assert sum(e - s for s, e in mine) <= sharding.local_rows(n, b, w, blk)
            seen += mine
        seen.sort()
        assert seen[0][0] == 0 and seen[-1][1] == n
        assert all(a[1] == c[0] for a, c in zip(seen, seen[1:]))
        assert sharding.local_rows(n, b, w, blk) * w >= n


class _CpuBackbone:
    """Stands in for the HIP encoder (no GPU in the CPU suite): features are a fixed function of the frame bytes."""
    def __call__(self, data):
        x = torch.from_numpy(np.ascontiguousarray(data)).float()
        x = x.reshape(x.shape[0], -1)
        return torch.stack([x.mean(1), x.std(1), x[:, 0], x[:, -1], x.abs().max(1).values], 1)


class _CpuNet:
    backbone = _CpuBackbone()


def _eval_worker(root, n_expected, q):
    """config C4's code path on gloo: tennis_amd.evaluate.save_features_sharded (the function evaluate.main drives for
    --save_feats --num_gpus N) -> per-rank .npy files + all-gathered (N, F) matrix; ranks formed by sharding.launch."""
    import numpy as np
    import torch.distributed as dist
    from tennis_amd import evaluate as ev, sharding
    from tennis_amd.dataset import DataLoader, TennisSet
    rank, world, dev = sharding.init_distributed()
    assert world == 2 and dev.type == "cpu"
    ds = TennisSet(root=root, split="test", model_id="c4", save_feats=True, frames_per_video=7, data_shape=16, split_first=2,
                   video_length=12)
    assert len(ds) == n_expected
    loader = DataLoader(ds, batch_size=4)
    stats = {}
    full, written = ev.save_features_sharded(_CpuNet(), loader, ds, device=dev, rank=rank, world=world, block=2, stats=stats)
    ref = torch.cat([_CpuNet.backbone(loader.collate(range(s, min(s + 4, len(ds))))[0]) for s in range(0, len(ds), 4)])
    ok = bool(torch.equal(full, ref))
    dist.barrier()
    on_disk = all(np.array_equal(np.load(ds.save_feature_path(i)), ref[i].numpy()) for i in range(len(ds)))
    q.put((rank, ok, on_disk, written, stats["frames_local"]))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_save_features_sharded_world2(tmp_path):
    from tennis_amd.dataset import TennisSet
    root = str(tmp_path)
    n = len(TennisSet(root=root, split="test", model_id="c4", save_feats=True, frames_per_video=7, data_shape=16, split_first=2,
                      video_length=12))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sharding.launch(_eval_worker, 2, (root, n, q))
    res = sorted(q.get(timeout=60) for _ in range(2))
    assert all(ok and disk for _, ok, disk, _, _ in res), res
    assert sum(w for *_, w, _ in res) == n                  # every frame written exactly once, by its owner
    assert sum(f for *_, f in res) == n and all(f > 0 for *_, f in res)


def test_bench_and_evaluate_self_launch_are_wired():
    """bench.py --gpus N / evaluate.py --num_gpus N started as plain python spawn their ranks (sharding.launch)
    instead of silently running one: the launch branch is taken when no launcher environment is present."""
    import inspect
    import bench
    from tennis_amd import evaluate as ev
    assert "sharding.launch(run, args.gpus" in inspect.getsource(bench.main)
    assert "sharding.launch(main, flags.num_gpus" in inspect.getsource(ev.main)
    assert not sharding.under_launcher()


def _dp_train_worker(rank, world, port, q):
    """Data-parallel recipe of tennis_amd.train.allreduce_and_step on CPU (gloo): each rank holds the gradient of the
    summed loss of ITS half of the batch (oracle), all-reduce SUM, update with rescale 1/global_batch -> the same
    parameters as one process on the whole batch."""
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from oracle import train_np as tn
    from tennis_amd import weights as W
    from tennis_amd.train import allreduce_and_step
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T, F, H, C_ = 8, 6, 16, 8, 11
    p = W.make_rnn_weights(1, "gru", F, H, "cnnrnn0_gru0_")
    p.update(W.make_dense_weights(2, C_, 2 * H, "cnnrnn0_dense0_"))
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (B, T, F)).astype(np.float32)
    y = rng.integers(0, C_, B)
    keys = sorted(p)
    sl = slice(rank * B // world, (rank + 1) * B // world)

    class FakeHead:                       # the trainer surface allreduce_and_step uses: .grads (flat tensor), .step()
        def __init__(self):
            _, _, g = tn.forward_backward(x[sl], y[sl], p)
            self.grads = torch.from_numpy(np.concatenate([g[k].ravel() for k in keys]))
            self.new = None

        def step(self, batch_size, lr, momentum, wd):
            g, off = {}, 0
            flat = self.grads.numpy()
            for k in keys:
                g[k] = flat[off:off + p[k].size].reshape(p[k].shape); off += p[k].size
            self.new, _ = tn.sgd_momentum({k: v.astype(np.float64) for k, v in p.items()}, g, {}, lr, momentum, wd,
                                          1.0 / batch_size)
    h = FakeHead()
    allreduce_and_step(h, B, 0.01, 0.9, 1e-4)
    _, _, gfull = tn.forward_backward(x, y, p)
    ref, _ = tn.sgd_momentum({k: v.astype(np.float64) for k, v in p.items()}, gfull, {}, 0.01, 0.9, 1e-4, 1.0 / B)
    q.put((rank, max(float(np.abs(h.new[k] - ref[k]).max()) for k in keys)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_step_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 411
    procs = [ctx.Process(target=_dp_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=120) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
    assert all(e < 1e-12 for _, e in res), res


class _LateEncoder:
    """Stands in for the pipelined HIP encoder: a forward only QUEUES its work; the output buffer is written when the caller
    joins (``join(1)``: everything but the newest call, ``join(0)``: everything) - so a loop that gathers a buffer before
    joining its forward, or reuses a buffer too early, reads stale rows and the test sees it."""

    def __init__(self, rank):
        self.rank, self.queue, self.calls, self.pipelined = rank, [], 0, False

    def set_pipelined(self, on):
        self.pipelined = on

    def __call__(self, x, out):
        i = self.calls
        self.calls += 1

        def work():
            out.copy_(x * 0 + (1000.0 * self.rank + i))
        if self.pipelined:
            out.fill_(-1.0)          # what a consumer would see if it did not wait
            self.queue.append(work)
        else:
            work()

    def join(self, lag):
        keep = self.queue[len(self.queue) - lag:] if lag else []
        for w in self.queue[:len(self.queue) - lag]:
            w()
        self.queue = keep


class _RecordingComm:
    """GroupComm (the torch.distributed transport bench.py falls back to) that keeps what every collective delivered."""

    def __init__(self):
        from tennis_amd.comm import GroupComm
        self.inner, self.log = GroupComm(), []

    def allgather_features(self, shard, out):
        h = self.inner.allgather_features(shard, out)
        h.wait()
        self.log.append(out.clone())
        return h


def _steploop_worker(rank, world, port, pipelined, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    b, f = 3, 4
    enc = _LateEncoder(rank)
    enc.set_pipelined(pipelined)
    x = torch.zeros((b, f))
    feats = [torch.empty((b, f)) for _ in range(2)]
    gathered = [torch.empty((world * b, f)) for _ in range(2)]
    comm = _RecordingComm()
    loop = bench.StepLoop(enc, x, feats, gathered, comm, world, pipelined)
    for i in range(k):
        loop.step(i)
    loop.drain(k)
    dist.barrier()
    ret[rank] = [t[:, 0].tolist() for t in comm.log]
    dist.destroy_process_group()


def test_bench_step_loop_world2():
    """bench.py's timed loop (StepLoop: forward, join one step behind, all-gather, drain) on gloo, world 2, through the
    torch.distributed transport: every one of the K steps is gathered exactly once, in order, with every rank's rows of THAT
    step - pipelined and joined, odd and even K (`bench.py --gpus 2` was only reachable on a multi-GPU box)."""
    port = 29561
    for pipelined in (True, False):
        for k in (1, 4, 5):
            with mp.Manager() as mgr:
                ret = mgr.dict()
                mp.spawn(_steploop_worker, args=(2, port, pipelined, k, ret), nprocs=2, join=True)
                port += 1
                want = [[1000.0 * r + i for r in range(2) for _ in range(3)] for i in range(k)]
                for r in range(2):
                    assert ret[r] == want, (pipelined, k, ret[r])


class _StandInBackbone:
    """What ``evaluate.extract_corpus`` needs of a backbone: frames (n, s, s, 3) uint8 -> (n, F) float32, a function of the frame's
    pixels only (so the gathered matrix cannot depend on which rank encoded a frame)."""

    def __call__(self, x):
        v = x.reshape(x.shape[0], -1).to(torch.float64)
        cols = [v.mean(1), v[:, ::7].sum(1) * 1e-3, v.std(1), v[:, 5], v.max(1).values]
        return torch.stack(cols, 1).to(torch.float32)


def _corpus_worker(rank, world, port, n_frames, batch, ret):
    from tennis_amd import evaluate as ev
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    full, st = ev.extract_corpus(_StandInBackbone(), n_frames, batch, 8, torch.device("cpu"), rank, world, block=2)
    ret[(world, rank)] = (full.numpy(), float(full.double().sum().item()), st["rounds"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_corpus_extract_world2_reproduces_the_world1_checksum():
    """``evaluate --corpus_frames N`` (BASELINE config C4) on two ranks gathers the matrix - and therefore the
    checksum - that one rank computes: frames are synthesised from their global index, batches are dealt to ranks in blocks,
    the chunked all-gather puts every block back at its rows (ragged last batch, uneven number of blocks per rank)."""
    n_frames, batch = 157, 16                 # 10 batches, the last one of 13 frames; 5 blocks of 2 batches over 2 ranks
    with mp.Manager() as mgr:
        ret = mgr.dict()
        _corpus_worker(0, 1, 0, n_frames, batch, ret)
        mp.spawn(_corpus_worker, args=(2, 29547, n_frames, batch, ret), nprocs=2, join=True)
        one, cs1, _ = ret[(1, 0)]
        assert one.shape == (n_frames, 5) and np.isfinite(one).all()
        for r in (0, 1):
            full, cs, rounds = ret[(2, r)]
            assert np.array_equal(full, one) and cs == cs1 and rounds >= 2


def test_corpus_extract_world4_ragged_last_round():
    """the same on FOUR ranks with a ragged last round - 11 batches in blocks of 2 are 6 blocks over 4 ranks: the
    second round has two ranks with a block (one of them the corpus' 13-frame tail in a half-empty block) and two with none, which
    still take part in the round's collective with padding rows.  Every rank ends with the world-1 matrix and checksum."""
    n_frames, batch = 173, 16                 # 11 batches (the last one of 13 frames) = 6 blocks of 2 batches (the last block holds one)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        _corpus_worker(0, 1, 0, n_frames, batch, ret)
        mp.spawn(_corpus_worker, args=(4, 29571, n_frames, batch, ret), nprocs=4, join=True)
        one, cs1, _ = ret[(1, 0)]
        assert one.shape == (n_frames, 5)
        for r in range(4):
            full, cs, rounds = ret[(4, r)]
            assert np.array_equal(full, one) and cs == cs1 and rounds == 2, (r, rounds)
