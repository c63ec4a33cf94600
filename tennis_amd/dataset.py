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

Not kept: JPEG decoding / split files / annotations (the 217 GB TenniSet frames
are not available; SURVEY §8f-3 lists the real input side as 'next').  Frames are
generated deterministically from (video, frame) so every rank and the oracle see
the same pixels.
"""
from __future__ import annotations

import math
import os
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
    def __init__(self, root="data", captions=False, transform=None, split="train", every=1, balance=True,
                 padding=1, stride=1, window=1, model_id="0000", split_id="02", flow=False, max_cap_len=-1,
                 vocab=None, inference=False, feats_model=None, save_feats=False,
                 # synthetic-source knobs (not in the reference):
                 data_shape=224, videos=("V006", "V007"), frames_per_video=16, seed=1234, split_first=0,
                 video_length=None):
        if captions:
            raise NotImplementedError("caption mode (dataset.py:154-183) is served by tennis_amd.captioning")
        if flow:
            raise NotImplementedError("optical-flow input is outside the accelerated hot path (SURVEY §2a)")
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

        self._frames_dir = os.path.join(root, "frames")
        self.output_dir = os.path.join(root, "outputs", model_id, split)
        self._load_feats = False
        self.feat_dir = os.path.join(root, "features", model_id)          # dataset.py:41
        if feats_model is not None:                                        # dataset.py:42-44
            self.feat_dir = os.path.join(root, "features", feats_model)
            self._load_feats = True

        self.classes = list(CLASSES)
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
        """HWC uint8 RGB 'decoded JPEG' for (video, frame), deterministic."""
        s = zlib.crc32(f"{video}:{frame}:{self._seed}".encode())
        return np.random.default_rng(s).integers(0, 256, (self._data_shape, self._data_shape, 3), dtype=np.uint8)

    def _load(self, video, frame):
        if self._load_feats:
            return np.load(self.get_feature_path(self.feat_dir, video, frame)).astype(np.float32)
        return self._transform(self.frame_u8(video, frame))

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

    def __init__(self, dataset, batch_size, shuffle=False, num_workers=0, last_batch="keep"):
        assert not shuffle, "the accelerated inference path is deterministic and un-shuffled"
        self.dataset, self.batch_size = dataset, batch_size

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        for s in range(0, n, self.batch_size):
            items = [self.dataset[i] for i in range(s, min(n, s + self.batch_size))]
            yield (np.stack([it[0] for it in items]), np.array([it[1] for it in items], dtype=np.float32),
                   np.array([it[2] for it in items], dtype=np.int64))
