"""Tennis video classification dataset — mirror of reference dataset.py::TennisSet
for the non-caption (frame / window) branch, over a SYNTHETIC source.

Kept from the reference: constructor signature (dataset.py:17-19), the sample
tuple ``(video_name, frame_number, class_name)``, the path scheme of
``get_image_path / get_feature_path / save_feature_path`` (dataset.py:135-150),
the window sampling of ``__getitem__`` (dataset.py:190-217: offsets
``range(int(-w/2), ceil(w/2))``, frame clamped to ``[0, max_frame]`` with
``max_frame`` snapped to an ``every`` multiple), the ±255 'OTH' padding frames
that ``save_feats=True`` adds per video (dataset.py:333-345) and the
``(img, label, idx)`` return with the same shapes: (3,H,W) / (T,3,H,W) float32
frames or (F,) / (T,F) features loaded from ``.npy``.

Two sources.  ON DISK, used when ``<root>/splits/<split_id>/<split>.txt`` exists: the
reference's layout (data/README.md) — split lines ``video frame``, per-video label files
``annotations/labels/<video>.txt`` (``frame class``), JPEG frames under ``frames/<video>.mp4/<chunk>/<frame>.jpg``
decoded on the host with Pillow (the reference decodes on the host too: ``mx.image.imread``) or, with
``decode="device"``, on the GPU by ``tennis_amd.image`` (one call per batch, bit-identical pixels), events = runs of equal labels,
video lengths from the frame directories, optional ``annotations/points.txt`` + ``captions.txt``, ``save_feats``
padding and ``_balance_classes`` (dataset.py:268-287,300-452).  Frames missing on disk are ignored, as in the
reference's second pass; extracting them from the ``.mp4`` is not done here.  SYNTHETIC otherwise (the 217 GB
TenniSet frames are not available): frames are generated deterministically from (video, frame) so every rank and
the oracle see the same pixels.

A transform with ``device_batched = True`` (``tennis_amd.transforms.Compose``) is not applied per frame:
``__getitem__`` returns the decoded uint8 frame and ``DataLoader`` runs the transform once per batch on the GPU.
"""
from __future__ import annotations

import logging
import math
import os
import random
import zlib

import numpy as np

CLASSES = ["OTH", "SFI", "SFF", "SFL", "SNI", "SNF", "SNL", "HFL", "HFR", "HNL", "HNR"]  # data/classes.names
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def default_transform(img_hwc_u8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize of the reference test transform (evaluate.py:96-97);
    Resize/CenterCrop are identities for frames synthesised at data_shape."""
    x = img_hwc_u8.astype(np.float32) / 255.0
    return np.ascontiguousarray(((x - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1))


class TennisSet:
    def __new__(cls, root="data", captions=False, *args, **kwargs):
        """``TennisSet(captions=True, ...)`` (reference dataset.py:17-19,52-74,154-183; train_gnmt.py:196-203) IS the caption-mode
        dataset: one sample per point - the clip's frame features and the caption's token ids.  It is served by
        ``tennis_amd.captions.CaptionSet`` (same arguments, same sample tuple); the frame / window branch is this class."""
        if captions and cls is TennisSet:
            from .captions import CaptionSet
            # (positional arguments behind `captions`, in __init__'s own order: derived, so that the two cannot drift - ADVICE r5)
            import inspect
            names = tuple(inspect.signature(TennisSet.__init__).parameters)[3:]
            kw = dict(zip(names, args))
            kw.update(kwargs)
            if kw.get("flow"):
                raise NotImplementedError("optical-flow input is outside the accelerated hot path (SURVEY §2a)")
            return CaptionSet(split=kw.get("split", "train"), every=kw.get("every", 1), max_cap_len=kw.get("max_cap_len", -1),
                              vocab=kw.get("vocab"), inference=kw.get("inference", False), root=root,
                              split_id=kw.get("split_id", "02"), feats_model=kw.get("feats_model"),
                              **{k: kw[k] for k in ("n_points", "feature_dim", "mean_frames", "seed") if k in kw})
        return super().__new__(cls)

    def __init__(self, root="data", captions=False, transform=None, split="train", every=1, balance=True,
                 padding=1, stride=1, window=1, model_id="0000", split_id="02", flow=False, max_cap_len=-1,
                 vocab=None, inference=False, feats_model=None, save_feats=False,
                 # synthetic-source knobs (not in the reference):
                 data_shape=224, videos=("V006", "V007"), frames_per_video=16, seed=1234, split_first=0,
                 video_length=None, decode="host"):
        if flow:
            raise NotImplementedError("optical-flow input is outside the accelerated hot path (SURVEY §2a)")
        if captions:      # __new__ redirects TennisSet(captions=True) to CaptionSet and never gets here; a subclass / object.__new__ path does
            raise NotImplementedError("caption mode is served by tennis_amd.captions.CaptionSet: construct TennisSet(captions=True, ...) "
                                      "itself (not a subclass), or CaptionSet directly")
        self._root = root
        self._captions = captions
        self._split = split
        self._balance = balance
        self._every = every
        self._padding = padding
        self._stride = stride
        self._window = window
        self._transform = transform if transform is not None else default_transform
        self._flow = flow
        self._inference = inference
        self._save_feats = save_feats
        self._data_shape = data_shape
        self._seed = seed
        # where the JPEG frames of an on-disk dataset are decoded: "host" (Pillow, as the reference's mx.image.imread on
        # its DataLoader workers), "device" (tennis_amd.image: the files' bytes go to the GPU, one decode per batch; files
        # the device decoder refuses raise), "auto" (device, and a batch it refuses is decoded on the host instead)
        if decode not in ("host", "device", "auto"):
            raise ValueError("decode must be 'host', 'device' or 'auto'")
        self.decode = decode

        self._frames_dir = os.path.join(root, "frames")
        self.output_dir = os.path.join(root, "outputs", model_id, split)
        self._load_feats = False
        self.feat_dir = os.path.join(root, "features", model_id)          # dataset.py:41
        if feats_model is not None:                                        # dataset.py:42-44
            self.feat_dir = os.path.join(root, "features", feats_model)
            self._load_feats = True

        self.classes = self._get_classes(root)
        self._splits_dir = os.path.join(root, "splits")
        self._annotations_dir = os.path.join(root, "annotations")
        self._labels_dir = os.path.join(root, "annotations", "labels")
        self._events, self._points, self._videos = [], {}, list(videos)
        self.on_disk = os.path.exists(os.path.join(self._splits_dir, split_id, split + ".txt"))
        if self.on_disk:
            self._samples, self._videos, self._events, self._points = self.load_data(split_id)
            self._video_lengths = self._get_video_lengths()
        else:
            self._synthesise(videos, frames_per_video, split_first, video_length, every, save_feats, seed)
        if self._balance and self.on_disk:                                 # dataset.py:74-75 (synthetic labels are uniform)
            self._samples = self._balance_classes()

    def _synthesise(self, videos, frames_per_video, split_first, video_length, every, save_feats, seed):
        vlen = video_length if video_length is not None else split_first + frames_per_video
        self._video_lengths = {v: vlen for v in videos}
        rng = np.random.default_rng(seed)
        self._samples = []
        pads = []
        for v in videos:
            frames = list(range(split_first, split_first + frames_per_video, every))
            labels = rng.integers(0, len(self.classes), len(frames))
            for f, l in zip(frames, labels):
                self._samples.append([v, f, self.classes[int(l)]])
            if save_feats:
                # dataset.py:333-345: +-255 frames around the split's range, labelled OTH, so windowed
                # models can read features past the split boundary; frames that do not exist in the
                # video are ignored (dataset.py:347-372)
                min_f, max_f = frames[0], frames[-1]
                for i in range(1, 256):
                    for f in (min_f - i, max_f + i):
                        if 0 <= f < vlen:
                            pads.append([v, f, "OTH"])
        self._samples += pads

    # ---- on-disk source (reference dataset.py:300-452) ------------------------------
    @staticmethod
    def _get_classes(root="data"):                                          # dataset.py:249-261
        names_file = os.path.join(root, "classes.names")
        if os.path.exists(names_file):
            with open(names_file) as f:
                return [line.strip() for line in f if line.strip()]
        return list(CLASSES)

    def load_data(self, split_id="01"):
        """-> samples [[video, frame, class]], videos, events [[video, first, last, class]], points {id: [...]}"""
        with open(os.path.join(self._splits_dir, split_id, self._split + ".txt")) as f:
            samples = [[ln.split()[0], int(ln.split()[1])] for ln in f if ln.strip()]
        videos = list({s[0] for s in samples})
        labels = {v: {} for v in videos}
        if self._save_feats:                                               # :333-345
            for v in videos:
                fr = [s[1] for s in samples if s[0] == v]
                lo, hi = min(fr), max(fr)
                for i in range(1, 256):
                    samples += [[v, lo - i], [v, hi + i]]
                    labels[v][lo - i] = labels[v][hi + i] = "OTH"
        kept = []
        for s in samples:                                                  # :347-372, second pass: ignore
            if os.path.exists(self.get_image_path(self._frames_dir, s[0], s[1])):
                kept.append(s)
            else:
                logging.info("%s does not exist, will ignore sample.", self.get_image_path(self._frames_dir, s[0], s[1]))
        samples = kept
        for v in videos:                                                   # :377-383 (the files override the padding's OTH)
            with open(os.path.join(self._labels_dir, v + ".txt")) as f:
                for ln in f:
                    parts = ln.split()
                    if parts:
                        labels[v][int(parts[0])] = parts[1]
        in_set = {v: [] for v in videos}
        for s in samples:                                                  # :391-393
            s.append(labels[s[0]][s[1]])
            in_set[s[0]].append(s[1])
        events = []
        for v in in_set:                                                   # :396-409: runs of one class; the first run is
            cur, start, last = "OTH", -1, -1                               # emitted as OTH even when it is empty
            for fr in sorted(in_set[v]):
                if start < 0:
                    start = last = fr
                if labels[v][fr] != cur:
                    events.append([v, start, last, cur])
                    cur, start = labels[v][fr], fr
                last = fr
            events.append([v, start, last, cur])
        points = {}
        pts_file, cap_file = (os.path.join(self._annotations_dir, n) for n in ("points.txt", "captions.txt"))
        if os.path.exists(pts_file) and os.path.exists(cap_file):          # :411-433
            with open(pts_file) as f:
                pts = [ln.split() for ln in f if ln.strip()]
            with open(cap_file) as f:
                caps = dict(ln.rstrip("\n").split("\t")[:2] for ln in f if ln.strip())
            for p in pts:
                if p[1] in videos and int(p[2]) in in_set[p[1]]:
                    points[p[0]] = p[1:] + [caps[p[0]]]
        return samples, videos, events, points

    def _get_video_lengths(self):                                          # :438-452: name of the last frame file
        lengths = {}
        for s in self._samples:
            v = s[0]
            if v not in lengths:
                vdir = os.path.join(self._frames_dir, v + ".mp4")
                largest_dir = sorted(os.listdir(vdir))[-1]
                assert largest_dir.isdigit(), f"Expects the directory {vdir} to only contain numbered subdirs"
                lengths[v] = int(sorted(os.listdir(os.path.join(vdir, largest_dir)))[-1][:-4])
        return lengths

    def _balance_classes(self):
        """dataset.py:268-287: thin out 'OTH' to the size of the next most frequent class with uniform random
        sampling (python's ``random``; seed it for a reproducible subset)."""
        counts = self.class_counts()
        ratio = max(counts[1:]) / float(counts[0] + 1)
        return [s for s in self._samples if not (s[2] == "OTH" and random.uniform(0, 1) > ratio)]

    def stats(self):                                                       # dataset.py:96-131, frame branch
        frame_counts, event_counts = self.class_counts(), [0] * len(self.classes)
        for e in self._events:
            event_counts[self.classes.index(e[3])] += 1
        out = "Split: {}\n".format(self._split)
        out += "{0: <6} {1: <8} {2: <8} {3: <5}\n".format("Class", "# Frames", "# Events", "FperE")
        for i, c in enumerate(self.classes):
            out += "{0: <6} {1: <8} {2: <8} {3: <5}\n".format(c, frame_counts[i], event_counts[i],
                                                              int(frame_counts[i] / (event_counts[i] + .00001)))
        return out

    def __str__(self):
        return "\n\n" + self.__class__.__name__ + "\n" + self.stats() + "\n"

    # ---- reference helpers -------------------------------------------------
    def __len__(self):
        return len(self._samples)

    @property
    def num_class(self):
        return len(self.classes)

    @staticmethod
    def get_image_path(root_dir, video_name, frame_number, chunk_size=1000):           # dataset.py:135-138
        chunk = int(frame_number / chunk_size) * chunk_size
        return os.path.join(root_dir, video_name + ".mp4", "{:010d}".format(chunk),
                            "{:010d}.jpg".format(frame_number))

    @staticmethod
    def get_feature_path(feat_dir, video_name, frame_number, chunk_size=1000):         # dataset.py:140-143
        chunk = int(frame_number / chunk_size) * chunk_size
        return os.path.join(feat_dir, video_name + ".mp4", "{:010d}".format(chunk),
                            "{:010d}.npy".format(frame_number))

    def save_feature_path(self, idx, chunk_size=1000):                                 # dataset.py:145-150
        sample = self._samples[idx]
        return self.get_feature_path(self.feat_dir, sample[0], sample[1], chunk_size)

    def class_counts(self):
        counts = [0] * len(self.classes)
        for s in self._samples:
            counts[self.classes.index(s[2])] += 1
        return counts

    # ---- synthetic frame source ----------------------------------------------
    def frame_u8(self, video, frame) -> np.ndarray:
        """HWC uint8 RGB frame: the decoded JPEG (``mx.image.imread(path, 1)``, dataset.py:204,216) when the data
        is on disk, else a deterministic synthetic frame."""
        if self.on_disk:
            from PIL import Image
            with Image.open(self.get_image_path(self._frames_dir, video, frame)) as im:
                return np.asarray(im.convert("RGB"))
        s = zlib.crc32(f"{video}:{frame}:{self._seed}".encode())
        return np.random.default_rng(s).integers(0, 256, (self._data_shape, self._data_shape, 3), dtype=np.uint8)

    def frame_bytes(self, video, frame) -> bytes:
        """the frame's JPEG file as it is on disk (device decode route)"""
        with open(self.get_image_path(self._frames_dir, video, frame), "rb") as f:
            return f.read()

    def sample_frames(self, idx):
        """[(video, frame), ...] that item ``idx`` reads: one frame, or the frames of its window (dataset.py:190-201)"""
        sample = self._samples[idx]
        if self._window > 1:
            return [(sample[0], f) for f in self.window_frames(sample)]
        return [(sample[0], sample[1])]

    def _load(self, video, frame):
        if self._load_feats:
            return np.load(self.get_feature_path(self.feat_dir, video, frame)).astype(np.float32)
        img = self.frame_u8(video, frame)
        return img if getattr(self._transform, "device_batched", False) else self._transform(img)

    def window_frames(self, sample):
        """Frame numbers a window sample reads (dataset.py:190-201)."""
        offsets = list(range(int(-self._window / 2), int(math.ceil(self._window / 2))))
        max_frame = self._video_lengths[sample[0]] - self._every
        for i in range(self._every):
            if (max_frame - i) % self._every == 0:
                max_frame -= i
                break
        return [min(max(0, sample[1] + o * self._stride), int(max_frame)) for o in offsets]

    def __getitem__(self, idx):                                                        # dataset.py:184-233
        sample = self._samples[idx]
        label = self.classes.index(sample[2])
        if self._window > 1:
            img = np.stack([self._load(sample[0], f) for f in self.window_frames(sample)])
        else:
            img = self._load(sample[0], sample[1])
        return img, label, idx


class DataLoader:
    """``gluon.data.DataLoader(dataset, batch_size, shuffle=False)`` stand-in: yields
    (data, labels, idxs) numpy batches in order, last batch kept (evaluate.py:113)."""

    def __init__(self, dataset, batch_size, shuffle=False, num_workers=0, last_batch="keep", seed=0):
        """``shuffle=True`` (training, train.py:189): a fresh seeded permutation per epoch; ``last_batch='discard'`` drops a
        ragged final batch (the fine-tuning step needs a fixed batch for its BatchNorm statistics).  ``num_workers`` (the
        reference's DataLoader worker processes, train.py:101-102) > 0 on the device-decode route: that many host threads (at
        most 2), each with its own JPEG decoder on its own stream, read and decode the following batches while the caller works
        on the current one - one decoder's file reads, header walk and host-to-device copy run beside the other's kernels."""
        self.dataset, self.batch_size, self.shuffle, self.last_batch = dataset, batch_size, shuffle, last_batch
        self._rng = np.random.default_rng(seed)
        self.num_workers = max(0, min(int(num_workers), 2))
        self._tls = None
        self._decoders = []          # (stream, JpegDecoder) of the worker threads, kept across epochs

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.last_batch == "discard" else (n + self.batch_size - 1) // self.batch_size

    def _device_route(self):
        ds = self.dataset
        return (getattr(ds, "decode", "host") != "host" and getattr(ds, "on_disk", False) and not ds._load_feats
                and getattr(getattr(ds, "_transform", None), "device_batched", False))

    def _decode_on_device(self, ids, decoder=None):
        """the batch's JPEG files -> (N[, T], H, W, 3) uint8 on the GPU in one decode call, or None if the device decoder
        refused them and the dataset allows the host route instead (decode="auto")"""
        from . import image
        ds = self.dataset
        frames = [ds.sample_frames(int(i)) for i in ids]
        bufs = [ds.frame_bytes(v, f) for fr in frames for (v, f) in fr]
        try:
            rgb = decoder.decode(bufs) if decoder is not None else image.imdecode_batch(bufs)
        except image.UnsupportedJpeg:
            if ds.decode != "auto":
                raise
            return None
        return rgb.view(len(ids), ds._window, *rgb.shape[1:]) if ds._window > 1 else rgb

    def collate(self, ids, rgb=None):
        """(data, labels, idxs) for the dataset items ``ids`` (what one iteration step yields); ``rgb``: the batch's frames
        already decoded on the device by a worker thread."""
        ds = self.dataset
        tf = getattr(ds, "_transform", None)
        if self._device_route():
            if rgb is None:
                rgb = self._decode_on_device(ids)
            if rgb is not None:
                labels = np.array([ds.classes.index(ds._samples[int(i)][2]) for i in ids], dtype=np.float32)
                return tf(rgb), labels, np.array([int(i) for i in ids], dtype=np.int64)
        items = [self.dataset[int(i)] for i in ids]
        data = np.stack([it[0] for it in items])
        if getattr(tf, "device_batched", False) and not self.dataset._load_feats:
            data = tf(data)                                                # one Resize+CenterCrop launch per batch
        return (data, np.array([it[1] for it in items], dtype=np.float32),
                np.array([it[2] for it in items], dtype=np.int64))

    def _batches(self):
        n = len(self.dataset)
        order = self._rng.permutation(n) if self.shuffle else np.arange(n)
        for s in range(0, n, self.batch_size):
            ids = order[s:s + self.batch_size]
            if self.last_batch == "discard" and len(ids) < self.batch_size:
                break
            yield ids

    def batches_of(self, rank: int, world: int):
        """The batches b with b mod world == rank, collated: a rank of a sharded run reads, decodes and resizes only its own."""
        for b, ids in enumerate(self._batches()):
            if b % world == rank:
                yield self.collate(ids)

    def __iter__(self):
        if self.num_workers == 0 or not self._device_route():
            for ids in self._batches():
                yield self.collate(ids)
            return
        # worker threads: each owns a decoder on its own stream (tn_jpeg handles are not re-entrant); batches come out in order
        import threading
        from concurrent.futures import ThreadPoolExecutor

        import torch

        from . import _lib, image
        tls = threading.local()
        lock = threading.Lock()
        free = list(self._decoders)      # decoders (contexts, streams, pinned workspaces) of earlier epochs are re-used

        def work(ids):
            if not hasattr(tls, "dec"):
                with lock:
                    if free:
                        tls.stream, tls.dec = free.pop()
                    else:
                        tls.stream = torch.cuda.Stream()
                        tls.dec = image.JpegDecoder(_lib.Context(stream=tls.stream))
                        self._decoders.append((tls.stream, tls.dec))
            # the output is allocated AND written under the worker's stream: the caching allocator then never hands the worker
            # a block whose last reader is still queued on another stream (tn_jpeg_decode returns with its stream synchronised)
            with torch.cuda.stream(tls.stream):
                return self._decode_on_device(ids, tls.dec)

        with ThreadPoolExecutor(max_workers=self.num_workers) as pool:
            pending = []
            for ids in self._batches():
                pending.append((ids, pool.submit(work, ids)))
                if len(pending) > self.num_workers:
                    i0, fut = pending.pop(0)
                    yield self.collate(i0, fut.result())
            for i0, fut in pending:
                yield self.collate(i0, fut.result())
