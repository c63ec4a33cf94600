"""Evaluation / feature-extraction driver — mirror of reference evaluate.py for the
hot path: same flag names, ``save_features`` (evaluate.py:306-321) and
``evaluate_model`` (evaluate.py:274-303) with the same results.

Differences that are deliberate: one device per rank
(no serialised per-device inner loop, no ``idxs`` rebinding bug), one D2H copy per
batch instead of one per row, frames sharded by batch across ranks.

``--num_gpus N`` (reference evaluate.py:101-102: the ``ctx`` list each batch is split over) starts N processes of
this script, one per GPU (``sharding.launch``; under ``torch.distributed.run`` the ranks exist already):
``--save_feats`` then goes through ``save_features_sharded`` — every rank encodes its batches, writes their
``.npy`` files and the feature rows are all-gathered over RCCL so that each rank ends with the whole (N, F)
matrix; testing shards the batches and sums the confusion matrices.  ``--corpus_frames N`` runs BASELINE config
C4 — a synthetic N-frame corpus (786 455 = the 5-match corpus) through the same sharded encode +
chunked, overlapped all-gather — and prints one JSON line with the rates.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

from . import sharding, transforms
from .dataset import DataLoader, TennisSet
from .metrics.vision import PRF1
from .model_zoo import get_model
from .models.vision.definitions import CNNRNN, FrameModel, TemporalPooling


def evaluate_model(net, loader, dataset, metrics, ctx=None):
    """Reference evaluate.py:274-303: forward, metric.update, results[img_path] = raw
    logits, ground_truths[img_path] = class index."""
    results, ground_truths = dict(), dict()
    for data, labels, idxs in loader:
        outputs = net(data)
        lab = torch.from_numpy(labels).to(outputs.device)
        for metric in metrics:
            metric.update([lab], [outputs])
        out = outputs.cpu().numpy()
        for i, idx in enumerate(int(j) for j in idxs):
            sample = dataset._samples[idx]
            img_path = dataset.get_image_path(dataset._frames_dir, sample[0], sample[1])
            results[img_path] = out[i]
            ground_truths[img_path] = dataset.classes.index(sample[2])
    return results, ground_truths


class NpyWriter:
    """The ``.npy`` side of ``--save_feats``: ``tn_npy_writer_*`` (csrc/npy_host.hip), a pool of host threads behind the C ABI that
    writes one NumPy-format-1.0 float32 file per feature row, byte for byte what ``np.save(path, row)`` writes, creating the
    directories and skipping files that exist (reference evaluate.py:312-316).  ``submit`` returns as soon as the rows are copied:
    the files of batch i are written while the GPU encodes batch i + 1 (the reference's np.save loop costs 13 ms of Python per 256
    frames against 1.8 ms of GPU time for their features)."""

    def __init__(self, threads: int | None = None):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.tn_npy_writer_create(int(threads or min(16, os.cpu_count() or 4)), C.byref(h)), "tn_npy_writer_create")
        self.handle = h
        self._written = self._skipped = 0

    def submit(self, rows: np.ndarray, paths, skip_existing: bool = True):
        C = self._C
        rows = np.ascontiguousarray(rows, np.float32)
        assert rows.ndim == 2 and rows.shape[0] == len(paths)
        arr = (C.c_char_p * len(paths))(*[os.fsencode(p) for p in paths])
        self._lib.check(self.lib.tn_npy_writer_submit(self.handle, rows.ctypes.data_as(C.c_void_p), rows.shape[0], rows.shape[1], arr,
                                                      1 if skip_existing else 0), "tn_npy_writer_submit")

    @staticmethod
    def sweep_stale(root: str, older_than_s: float = 600.0) -> int:
        """A run that is killed mid-write leaves ``<frame>.npy.tmp.<pid>.<thread>`` files behind (never a truncated ``.npy``:
        files are published by link / rename of a finished temporary).  Removes those under ``root`` that nobody has touched for
        ``older_than_s`` seconds (a live writer's temporaries are younger); -> the number removed.  Called by ``save_features`` on
        the feature directory before it writes."""
        import time
        n, now = 0, time.time()
        for d, _dirs, files in os.walk(root):
            for f in files:
                if ".npy.tmp." in f:
                    p = os.path.join(d, f)
                    try:
                        if now - os.path.getmtime(p) > older_than_s:
                            os.unlink(p)
                            n += 1
                    except OSError:
                        pass
        return n

    def drain(self):
        """waits for everything submitted; -> (files written, files skipped) since the last drain"""
        C = self._C
        w, s = C.c_int64(), C.c_int64()
        self._lib.check(self.lib.tn_npy_writer_drain(self.handle, C.byref(w), C.byref(s)), "tn_npy_writer_drain")
        out = (w.value - self._written, s.value - self._skipped)
        self._written, self._skipped = w.value, s.value
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tn_npy_writer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def save_features(net, loader, dataset, ctx=None, verbose=True):
    """Reference evaluate.py:306-321: feat = net.backbone(x); one float32 ``.npy`` per
    frame at save_feature_path(idx), skipped when the file already exists.  The files are written by ``NpyWriter``'s threads
    behind the next batch's encode (round 4); the count returned is the number of files that did not exist."""
    writer = NpyWriter()
    if os.path.isdir(getattr(dataset, "feat_dir", "") or ""):
        NpyWriter.sweep_stale(dataset.feat_dir)
    pending = None          # (features on the device, paths) of the batch before: copied to the host and handed to the writer
                            # AFTER the next batch's forward has been queued, so that the copy waits for nothing but its own batch

    def flush(item):
        feat, paths = item
        if verbose:
            for feat_path in paths:
                if not os.path.exists(feat_path):
                    print("Saving %s" % feat_path)
        writer.submit(feat.cpu().numpy(), paths)

    try:
        for data, _labels, idxs in loader:
            feat = net.backbone(data)
            if pending is not None:
                flush(pending)
            pending = (feat, [dataset.save_feature_path(int(j)) for j in idxs])
        if pending is not None:
            flush(pending)
        written, _skipped = writer.drain()
    finally:
        writer.close()
    return written


def _backbone_dim(net, loader):
    """Feature width F of ``net.backbone`` (1024 at 224x224, 4096 at 512x512): asked of the engine when it exists,
    else measured on the first sample."""
    eng = getattr(net.backbone, "_engine", None)
    if eng is not None and hasattr(eng, "feature_dim"):
        return int(eng.feature_dim)
    data, _, _ = loader.collate([0])
    return int(net.backbone(data).shape[-1])


def save_features_sharded(net, loader, dataset, device=None, rank=None, world=None, group=None, block=1, verbose=False,
                          write=True, stats=None):
    """``save_features`` (evaluate.py:306-321) for N ranks: rank r encodes the batches ``sharding.rank_batches`` gives
    it (dataset order, no shuffle), writes their ``.npy`` files (skip-if-exists, as the reference) and the rows are
    all-gathered round by round behind the compute.  Returns (features (len(dataset), F) in dataset order — identical
    on every rank —, files written by this rank)."""
    if rank is None or world is None:
        rank, world = sharding._rank(group), sharding._world(group)
    fdim = _backbone_dim(net, loader)
    writer = NpyWriter() if write else None
    pending = []            # the batch before this one: written out once this one's forward is queued (see save_features)

    def flush():
        while pending:
            feat, paths = pending.pop(0)
            if verbose:
                for feat_path in paths:
                    if not os.path.exists(feat_path):
                        print("Saving %s" % feat_path)
            writer.submit(feat.detach().cpu().numpy(), paths)

    def encode(s, e):
        data, _labels, idxs = loader.collate(range(s, e))
        feat = net.backbone(data)
        if write:
            flush()
            pending.append((feat, [dataset.save_feature_path(int(j)) for j in idxs]))
        return feat
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    comm = sharding.feature_comm(device, group) if world > 1 else None
    try:
        full = sharding.extract_features_sharded(encode, len(dataset), loader.batch_size, fdim, device, rank=rank, world=world,
                                                 group=group, block=block, stats=stats, comm=comm)
        if writer is not None:
            flush()
        written = writer.drain()[0] if writer is not None else 0
    finally:
        if writer is not None:
            writer.close()
    return full, written


class SyntheticCorpus:
    """A corpus of ``n_frames`` synthetic decoded frames generated ON the device batch by batch (uint8 NHWC, seeded by
    the global frame index range, so every rank produces the same frame for the same index): stands in for the
    786 455 frames of the 5-match corpus, which cannot be shipped."""

    def __init__(self, n_frames, size, device, seed=1234):
        self.n, self.size, self.device, self.seed = int(n_frames), int(size), device, seed

    def __len__(self):
        return self.n

    def frames(self, s, e):
        g = torch.Generator(device=self.device)
        g.manual_seed(self.seed + s)
        return torch.randint(0, 256, (e - s, self.size, self.size, 3), generator=g, device=self.device, dtype=torch.uint8)


def extract_corpus(backbone, n_frames, batch, size, device, rank, world, block=4, seed=1234, reuse_frames=False):
    """BASELINE config C4: feature-extract over an ``n_frames`` corpus sharded over the ranks with the RCCL all-gather
    of the rows (chunked: one collective per round of ``block`` batches per rank, in flight behind the next round).
    Every batch of frames is generated on the device from its frame indices inside the loop (the result is then the
    same for any number of ranks: the checksum of the gathered matrix is a parity check); ``reuse_frames`` generates
    a rank's first batch once and re-uses it (frame synthesis is not part of the path).
    -> (features (n_frames, F) on every rank, dict of timings)."""
    corpus = SyntheticCorpus(n_frames, size, device, seed)
    cached = {}

    def encode_input(s, e):
        if reuse_frames:
            if (e - s) not in cached:
                cached[e - s] = corpus.frames(s, e)
            return cached[e - s]
        return corpus.frames(s, e)

    def encode(s, e):
        return backbone(encode_input(s, e))
    fdim = int(backbone(corpus.frames(0, min(batch, n_frames))).shape[-1])      # also builds the engine (untimed)
    # the HIP encoder runs consecutive batches pipelined (its two half-batch streams are joined once per gather round, not
    # in every forward); a batch's frames stay referenced until that join
    eng = getattr(backbone, "_engine", None)
    pipelined = eng is not None and hasattr(eng, "set_pipelined") and device.type == "cuda"
    live = []

    def encode_into(s, e, rows):
        x = encode_input(s, e)
        if x.shape[0] <= eng.max_batch:
            live.append(x)
            eng(x, out=rows)
        else:
            rows.copy_(backbone(x))

    def join():
        eng.join(0)
        eng.join(1)
        live.clear()
    fence = (lambda: (torch.distributed.barrier() if world > 1 else None, torch.cuda.synchronize() if device.type == "cuda" else None))
    comm = sharding.feature_comm(device) if world > 1 else None
    stats = {}
    fence()
    t0 = time.perf_counter()
    if pipelined:
        eng.set_pipelined(True)
    try:
        full = sharding.extract_features_sharded(encode, n_frames, batch, fdim, device, rank=rank, world=world, block=block, stats=stats,
                                                 encode_into=encode_into if pipelined else None, join=join if pipelined else None,
                                                 comm=comm)
    finally:
        if pipelined:
            eng.set_pipelined(False)
    fence()
    dt = time.perf_counter() - t0
    stats.update(seconds=dt, frames_per_sec=n_frames / dt, feature_dim=fdim,
                 gather_GBps_per_rank=(stats["gather_bytes_per_rank"] * max(world - 1, 0) / dt / 1e9))
    if device.type == "cuda":        # HBM in use on this rank's device when the corpus is done (everything: the library's workspace, torch's pool, the feature matrix)
        free_b, total_b = torch.cuda.mem_get_info(device)
        stats.update(hbm_used_GB=(total_b - free_b) / 1e9, hbm_total_GB=total_b / 1e9)
    return full, stats


def best_or_newest_params(mod_path, need_scores):
    """The parameter file the reference would load (evaluate.py:186-200,206-212,223-240): epoch with the best
    ``scores.txt`` line; without a scores file the newest ``NNNN.params`` (None when the directory holds neither and
    ``need_scores`` is False)."""
    from .train import best_epoch_from_scores, newest_params
    scores = os.path.join(mod_path, "scores.txt")
    if os.path.exists(scores):
        ep, sc = best_epoch_from_scores(scores)
        if ep >= 0:
            print("Testing best model from Epoch %d with score of %f" % (ep, sc))
            return os.path.join(mod_path, "{:04d}.params".format(ep))
    if need_scores and not os.path.isdir(mod_path):
        return None
    return newest_params(mod_path)


def build_parser():
    p = argparse.ArgumentParser(description="tennis_amd evaluate (flags of reference evaluate.py:30-75)")
    p.add_argument("--backbone", default="DenseNet121")
    p.add_argument("--fp16_conversion", default="nearest", choices=["nearest", "calibrated", "exact"],
                   help="how the checkpoint's fp32 conv weights become the fp16 model (not a reference flag): plain rounding, "
                        "rounding calibrated on the library's built-in calibration frames (the same model on every rank; features within "
                        "1e-3 of the fp32 evaluation on natural content at full speed, DESIGN.md), or hi + lo weight pairs (the bar "
                        "on any input at twice the MFMAs)")
    p.add_argument("--model_id", default="0000")
    p.add_argument("--split_id", default="02")
    p.add_argument("--split", default="test")
    p.add_argument("--data_shape", type=int, default=224)
    p.add_argument("--every", default="1, 1, 1")
    p.add_argument("--window", type=int, default=1)
    p.add_argument("--padding", type=int, default=1)
    p.add_argument("--stride", type=int, default=1)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--save_feats", action="store_true")
    p.add_argument("--feats_model", default=None)
    p.add_argument("--temp_pool", default=None, help="mean, max, gru or lstm")
    # the rest of the reference's flags (evaluate.py:30-75), so that its documented command lines parse unchanged, e.g.
    # `python evaluate.py --model_id 0042 --backbone DenseNet121 --temp_pool gru --window 30 --backbone_from_id 0006 --feats_model 0006
    # --freeze_backbone` (models/README.md:58)
    p.add_argument("--backbone_from_id", default=None,
                   help="load the frame model's parameters from the newest .params of this model id before the temporal model is put around it (evaluate.py:142-151)")
    p.add_argument("--freeze_backbone", action="store_true", help="accepted (evaluate.py:153-155 sets grad_req = 'null': nothing to do without gradients)")
    p.add_argument("--balance", default="True, False, False", help="accepted; the test set is never balanced (evaluate.py:109)")
    p.add_argument("--vis", action="store_true", help="visualisation output is outside the accelerated path (SURVEY 2a): refused")
    p.add_argument("--flow", default="", help="optical-flow input is outside the accelerated path (SURVEY 2a): anything but '' is refused")
    p.add_argument("--root", default="data")
    p.add_argument("--num_workers", type=int, default=3,
                   help="loader threads (reference: DataLoader worker processes, evaluate.py:113); on the device-decode route each owns a JPEG decoder on its own stream")
    p.add_argument("--decode", default="device", choices=["device", "host", "auto"],
                   help="where on-disk JPEG frames are decoded: on the GPU (tennis_amd.image), on the host (Pillow), or device with a host fallback for files the device decoder refuses")
    p.add_argument("--frames_per_video", type=int, default=16)
    p.add_argument("--exp_root", default=os.path.join("models", "vision", "experiments"))
    p.add_argument("--corpus_frames", type=int, default=0,
                   help="config C4: encode a synthetic corpus of this many frames sharded over --num_gpus ranks with "
                        "the all-gather of the feature rows (786455 = the 5-match corpus); prints one JSON line")
    p.add_argument("--gather_block", type=int, default=4, help="batches per rank per all-gather round")
    p.add_argument("--corpus_reuse_frames", action="store_true", help="C4: generate one synthetic batch per rank and re-use it")
    return p


def main(argv=None):
    flags = build_parser().parse_args(argv)
    if flags.num_gpus > 1 and not sharding.under_launcher():
        # evaluate.py:101-102 builds ctx = [gpu(0) .. gpu(N-1)] in one process; here: one process per GPU
        if torch.cuda.is_available() and flags.num_gpus > torch.cuda.device_count():
            raise SystemExit(f"--num_gpus {flags.num_gpus} but only {torch.cuda.device_count()} GPUs are visible")
        sharding.launch(main, flags.num_gpus, (list(sys.argv[1:] if argv is None else argv),))
        return 0
    rank, world, dev = sharding.init_distributed()
    try:
        rc = _main_rank(flags, rank, world, dev)
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.barrier()          # success path only: a rank that raised must not park the others' collectives
        return rc
    finally:
        if world > 1 and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def _main_rank(flags, rank, world, dev):
    if flags.flow or flags.vis:
        raise NotImplementedError("--flow / --vis: optical-flow input and visualisation are outside the accelerated path (SURVEY 2a)")
    if flags.num_workers < 0:                                               # the reference's -1 = cpu_count() (evaluate.py:80-81)
        flags.num_workers = 3                                               # (files -> features peaks at three decoder threads: scripts/bench_pipeline.py, profiles/r05_c_*)
    every = [int(s) for s in flags.every.split(",")]
    if flags.corpus_frames > 0:                                             # BASELINE config C4
        backbone = get_model(flags.backbone, pretrained=True, max_batch=flags.batch_size, conversion=flags.fp16_conversion).features
        full, st = extract_corpus(backbone, flags.corpus_frames, flags.batch_size, flags.data_shape, dev, rank, world,
                                  block=flags.gather_block, reuse_frames=flags.corpus_reuse_frames)
        if rank == 0:
            print(json.dumps({"config": "C4 corpus feature-extract + all-gather", "frames": flags.corpus_frames,
                              "n_gpus": world, "batch": flags.batch_size, "rounds": st["rounds"],
                              "frames_per_sec": round(st["frames_per_sec"], 1), "seconds": round(st["seconds"], 3),
                              "feature_matrix_MB": round(full.numel() * 4 / 1e6, 1),
                              "gather_GBps_per_rank": round(st["gather_GBps_per_rank"], 3),
                              "hbm_used_GB": round(st.get("hbm_used_GB", 0.0), 2), "hbm_total_GB": round(st.get("hbm_total_GB", 0.0), 1),
                              "checksum": float(full.double().sum().item())}))
        return 0
    # evaluate.py:91-98: Resize(s + 32) / CenterCrop(s) / ToTensor / Normalize, here one GPU launch per batch; the
    # synthetic source already produces data_shape frames, so only data on disk goes through it
    transform_test = None
    if flags.feats_model is None and os.path.exists(os.path.join(flags.root, "splits", flags.split_id, flags.split + ".txt")):
        transform_test = transforms.Compose([transforms.Resize(flags.data_shape + 32),
                                             transforms.CenterCrop(flags.data_shape),
                                             transforms.ToTensor(),
                                             transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    test_set = TennisSet(root=flags.root, transform=transform_test, split=flags.split, every=every[2], padding=flags.padding,
                         stride=flags.stride, window=flags.window, model_id=flags.model_id,
                         split_id=flags.split_id, balance=False, feats_model=flags.feats_model,
                         save_feats=flags.save_feats, data_shape=flags.data_shape,
                         frames_per_video=flags.frames_per_video, decode=flags.decode)
    test_data = DataLoader(test_set, batch_size=flags.batch_size, shuffle=False, num_workers=flags.num_workers)

    model = None
    if flags.feats_model is None:                                           # evaluate.py:118-135
        backbone_net = get_model(flags.backbone, pretrained=True, conversion=flags.fp16_conversion).features
        model = FrameModel(backbone_net, len(test_set.classes))
    elif flags.temp_pool in ["max", "mean"]:                                # evaluate.py:136-138
        backbone_net = get_model(flags.backbone, pretrained=True, conversion=flags.fp16_conversion).features
        model = FrameModel(backbone_net, len(test_set.classes))
    if flags.window > 1:                                                    # evaluate.py:139-162
        if flags.backbone_from_id and model is not None:                    # evaluate.py:142-151
            bb_dir = os.path.join(flags.exp_root, flags.backbone_from_id)
            if os.path.isdir(bb_dir):
                files = sorted((f for f in os.listdir(bb_dir) if f.endswith(".params")), reverse=True)
                if files:
                    model.initialize()
                    model.load_parameters(os.path.join(bb_dir, files[0]))
                    print("Loaded backbone params: {}".format(os.path.join(bb_dir, files[0])))
        if flags.temp_pool in ["gru", "lstm"]:
            model = CNNRNN(model, num_classes=len(test_set.classes), type=flags.temp_pool, hidden_size=128)
        elif flags.temp_pool not in ["mean", "max"]:
            raise AssertionError("window > 1 needs --temp_pool (the 3-D rdnet backbone is out of scope)")
    model.initialize()
    model.hybridize()

    # trained parameters: the best epoch of scores.txt (evaluate.py:186-200,223-240), else the newest NNNN.params
    # (:206-212); an experiment directory without either leaves the (seeded) initial parameters, with a notice
    mod_path = os.path.join(flags.exp_root, flags.model_id)
    if flags.temp_pool in ["max", "mean"] and flags.feats_model is not None and not flags.save_feats:
        mod_path = os.path.join(flags.exp_root, flags.feats_model)          # evaluate.py:219-222
    params_file = best_or_newest_params(mod_path, need_scores=False)
    if params_file is not None:
        model.load_parameters(params_file)
        print("Loaded model params: {}".format(params_file))
    elif rank == 0:
        print("no trained parameters under %s: evaluating the initial parameters" % mod_path)

    if flags.save_feats:                                                    # evaluate.py:186-204
        # one code path for any number of ranks (world 1: no collective, the shard IS the matrix)
        full, n = save_features_sharded(model, test_data, test_set, device=dev, rank=rank, world=world,
                                        block=flags.gather_block, verbose=(world == 1))
        print("[rank %d of %d] wrote %d feature files under %s; feature matrix %s on every rank" %
              (rank, world, n, test_set.feat_dir, tuple(full.shape)))
        return 0

    if flags.temp_pool in ["max", "mean"]:                                  # evaluate.py:242-244
        model = TemporalPooling(model, pool=flags.temp_pool, num_classes=0, feats=flags.feats_model is not None)
    test_metrics = [PRF1(label_names=test_set.classes)]
    tic = time.time()
    if world > 1:                                                           # each rank tests batches rank::world
        results, gts = evaluate_model(model, _RankBatches(test_data, rank, world), test_set, test_metrics)
        comm = sharding.feature_comm(dev)
        for m in test_metrics:                                              # confusion counts add up over the ranks
            packed = torch.from_numpy(np.concatenate([m.mat.ravel(), m.scores.ravel()])).to(dev)
            if comm is not None and packed.dtype in (torch.float32, torch.int64):
                comm.allreduce_(packed).wait()                              # tn_allreduce_* of the C-ABI
            else:
                torch.distributed.all_reduce(packed)
            packed = packed.cpu().numpy()
            m.mat = packed[:m.mat.size].reshape(m.mat.shape)
            m.scores = packed[m.mat.size:].reshape(m.scores.shape)
        if rank != 0:
            return 0
    else:
        results, gts = evaluate_model(model, test_data, test_set, test_metrics)
    str_ = "Test set:"                                                       # evaluate.py:250-255
    for i in range(len(test_set.classes)):
        str_ += "\n"
        for j in range(len(test_set.classes)):
            str_ += str(test_metrics[0].mat[i, j]) + "\t"
    print(str_)
    str_ = "[Finished] "
    for res in test_metrics[0].get():
        str_ += ", Test_{}={:.3f}".format(res[0], res[1])
    str_ += "  # Samples: {}, Time Taken: {:.1f}".format(len(test_set), time.time() - tic)
    print(str_)
    return 0


class _RankBatches:
    """The batches b with b mod world == rank of a loader (dataset order)."""

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world

    def __iter__(self):
        # batch ids are filtered BEFORE collation where the loader allows it (tennis_amd.dataset.DataLoader.batches_of): a
        # rank then reads, decodes and resizes only its own batches
        if hasattr(self.loader, "batches_of"):
            yield from self.loader.batches_of(self.rank, self.world)
            return
        for b, batch in enumerate(self.loader):
            if b % self.world == self.rank:
                yield batch


if __name__ == "__main__":
    raise SystemExit(main())
