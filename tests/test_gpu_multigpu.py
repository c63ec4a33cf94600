"""-m gpu: the multi-GPU drivers (BASELINE config C4) on however many GPUs the box has (1 on the test box):
evaluate.py --save_feats --num_gpus N and --corpus_frames through sharding.extract_features_sharded, bench.py's
self-spawned ranks, and the loud failure when more ranks are asked for than GPUs exist."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=900):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out
    return json.loads(lines[-1])


def test_evaluate_save_feats_num_gpus(tmp_path, report):
    """evaluate.py --save_feats --num_gpus <all GPUs of the box>: every frame's .npy is written once, by its owner, and
    equals the un-sharded backbone output."""
    from tennis_amd.dataset import DataLoader, TennisSet
    from tennis_amd.model_zoo import get_model
    n = torch.cuda.device_count()
    root = str(tmp_path)
    r = _run(["-m", "tennis_amd.evaluate", "--root", root, "--model_id", "0c4", "--save_feats", "--frames_per_video", "5",
              "--batch_size", "4", "--num_gpus", str(n), "--gather_block", "2", "--exp_root", os.path.join(root, "exp")])
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("feature matrix") == n
    ds = TennisSet(root=root, split="test", model_id="0c4", save_feats=True, frames_per_video=5, data_shape=224)
    loader = DataLoader(ds, batch_size=4)
    backbone = get_model("DenseNet121", pretrained=True).features
    worst = 0.0
    for data, _, idxs in loader:
        ref = backbone(data).cpu().numpy()
        for i, idx in enumerate(int(j) for j in idxs):
            got = np.load(ds.save_feature_path(idx))
            assert got.dtype == np.float32 and got.shape == (1024,)
            worst = max(worst, float(np.abs(got - ref[i]).max()))
    report["save_feats_num_gpus_vs_direct"] = worst
    assert worst == 0.0          # same kernels, same batches of 4: bit-identical


def test_corpus_mode_checksum_and_rates(report):
    """config C4 scaled down (2 000 frames, same code path): the gathered matrix is the one a plain loop produces."""
    from tennis_amd import evaluate as ev
    from tennis_amd.model_zoo import get_model
    n = torch.cuda.device_count()
    r = _run(["-m", "tennis_amd.evaluate", "--corpus_frames", "2000", "--batch_size", "256", "--num_gpus", str(n),
              "--gather_block", "2"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = _json_line(r.stdout)
    assert line["frames"] == 2000 and line["n_gpus"] == n and line["frames_per_sec"] > 1000
    dev = torch.device("cuda", 0)
    backbone = get_model("DenseNet121", pretrained=True, max_batch=256).features
    corpus = ev.SyntheticCorpus(2000, 224, dev)
    total = 0.0
    for s in range(0, 2000, 256):
        total += float(backbone(corpus.frames(s, min(2000, s + 256))).double().sum().item())
    report["corpus_mode_frames_per_sec"] = line["frames_per_sec"]
    assert abs(total - line["checksum"]) <= 1e-6 * abs(total)


def test_bench_line_and_self_spawn():
    """bench.py started as plain python: --gpus 1 runs in-process; the spawn path (what --gpus N > 1 takes) is driven
    with one rank; more ranks than GPUs is a loud error, not a single-rank line."""
    r = _run(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 5 and line["roofline"]["bound"] == "mfma"
    import bench
    assert line["config"]["timing"].startswith(f"median of {bench.MIN_TOTAL_STEPS // 5} fenced regions of exactly 5 steps")
    code = ("import sys; sys.path.insert(0, %r); import bench; from tennis_amd import sharding; "
            "sharding.launch(bench.run, 1, (['--gpus', '1', '--steps', '5', '--warmup', '2', '--no-cpu-baseline', '--single-region'],))" % ROOT)
    r = _run(["-c", code])
    assert r.returncode == 0, r.stdout + r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 1
    too_many = torch.cuda.device_count() + 1
    r = _run(["bench.py", "--gpus", str(too_many), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r.returncode != 0 and "GPUs are visible" in (r.stdout + r.stderr)
    if torch.cuda.device_count() >= 2:
        r = _run(["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"])
        assert r.returncode == 0, r.stdout + r.stderr
        assert _json_line(r.stdout)["n_gpus"] == 2


# ---- the exchange step behind the C-ABI (tn_comm_* / tn_allgather_features / tn_allreduce_*) ----
def _comm_rank_main(rank, world, idfile, outfile):
    """one rank of the C-ABI communicator test (a spawned process per GPU)"""
    import time
    import torch
    torch.cuda.set_device(rank)
    from tennis_amd.comm import Comm
    if rank == 0:
        uid = Comm.new_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)          # the out-of-band channel of this test: a file
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        uid = open(idfile, "rb").read()
    comm = Comm(rank, world, uid, device=rank)
    rows, f = 96, 1024
    g = torch.Generator().manual_seed(100 + rank)
    shard = torch.rand((rows, f), generator=g).cuda()
    out = torch.zeros((world * rows, f), device="cuda")
    comm.allgather_features(shard, out).wait()
    grads = torch.full((1000,), float(rank + 1), device="cuda")
    comm.allreduce_(grads).wait()
    counts = torch.arange(121, dtype=torch.int64, device="cuda") * (rank + 1)
    comm.allreduce_(counts).wait()
    torch.cuda.synchronize()
    d = comm.describe()          # RCCL's own answers when world > 1 (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)
    assert d["world"] == world and d["rank"] == rank and d["device"] == rank and d["rccl"] == (world > 1), d
    torch.save({"out": out.cpu(), "grads": grads.cpu(), "counts": counts.cpu()}, outfile % rank)
    comm.close()


@pytest.mark.parametrize("force_rccl", [False, True])
def test_comm_single_rank_through_the_abi(force_rccl):
    """world 1: the all-gather is the identity and the all-reduces leave their buffers alone - without RCCL, and with
    TN_COMM_FORCE_RCCL through a real one-rank RCCL communicator (librccl opened by its SONAME: the instance PyTorch holds)."""
    from tennis_amd.comm import Comm
    comm = Comm(0, 1, Comm.new_unique_id() if force_rccl else None, force_rccl=force_rccl)
    shard = torch.rand((64, 1024), device="cuda")
    out = torch.zeros_like(shard)
    comm.allgather_features(shard, out).wait()
    g = torch.rand(777, device="cuda")
    g0 = g.clone()
    comm.allreduce_(g, average=True).wait()
    c = torch.arange(50, dtype=torch.int64, device="cuda")
    comm.allreduce_(c).wait()
    torch.cuda.synchronize()
    assert torch.equal(out, shard) and torch.equal(g, g0) and torch.equal(c, torch.arange(50, dtype=torch.int64, device="cuda"))
    # what the communicator says about itself (round 6: tn_comm_world / _rank / _device are RCCL's own answers - ncclCommCount,
    # ncclCommUserRank, ncclCommCuDevice - when RCCL is behind the handle; bench.py prints them per rank in config.comm)
    d = comm.describe()
    assert d["world"] == 1 and d["rank"] == 0 and d["rccl"] == bool(force_rccl) and d["device"] == torch.cuda.current_device(), d
    comm.close()


def test_comm_all_ranks_of_the_box(tmp_path):
    """One process per GPU of the box (1 on the test box: the spawn path with one rank; asserted against the un-sharded
    matrix for any number): tn_allgather_features puts every rank's rows in rank-major order on every rank,
    tn_allreduce_f32 / _i64 sum over the ranks."""
    import torch.multiprocessing as mp
    world = torch.cuda.device_count()
    idfile, outfile = str(tmp_path / "uid"), str(tmp_path / "rank%d.pt")
    mp.spawn(_comm_rank_main, args=(world, idfile, outfile), nprocs=world, join=True)
    rows, f = 96, 1024
    expect = torch.cat([torch.rand((rows, f), generator=torch.Generator().manual_seed(100 + r)) for r in range(world)])
    for r in range(world):
        got = torch.load(outfile % r)
        assert torch.equal(got["out"], expect)
        assert torch.equal(got["grads"], torch.full((1000,), float(world * (world + 1) // 2)))
        assert torch.equal(got["counts"], torch.arange(121, dtype=torch.int64) * (world * (world + 1) // 2))
