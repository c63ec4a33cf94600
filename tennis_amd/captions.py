"""Caption-mode dataset, batching and the captioning evaluate() driver — mirrors of the
caption branch of reference dataset.py::TennisSet (dataset.py:52-74,154-183,235-247),
utils/captioning.py::get_dataloaders/write_sentences (:28-95) and
train_gnmt.py::evaluate (:264-302).  Two sources: ON DISK when ``root`` holds the reference's layout
(``splits/<split_id>/<split>.txt``, ``annotations/points.txt`` + ``captions.txt``, per-frame ``.npy`` features under
``features/<feats_model>/`` as ``evaluate --save_feats`` writes them; parsed by ``TennisSet.load_data``), else SYNTHETIC
(the TenniSet features and captions are not available).

One sample = one *point*: every `every`-th frame feature in [start, end), stacked (T,F), plus
the caption ids ``[<bos>] + vocab[tokens][:max_cap_len] + [<eos>]`` as int32 (dataset.py:67-73);
val/test reuse the train vocab (train_gnmt.py:200-203).  Batches are zero-padded like
``gluonnlp.data.batchify.Pad()`` (target padding is id 0), lengths are float32.
"""
from __future__ import annotations

import io
import math
import zlib

import numpy as np
import torch

from .models.captioning.gnmt import Vocab

WORDS = ("the near far player serves hits a forehand backhand return volley into net in out left right "
         "middle ball wide deep cross court down line fault ace winner long short lob smash second first").split()


class CaptionSet:
    """``TennisSet(captions=True, feats_model=...)`` counterpart."""

    def __init__(self, split="train", every=1, max_cap_len=-1, vocab=None, inference=False, n_points=24,
                 feature_dim=1024, mean_frames=40, seed=7, root=None, split_id="02", feats_model=None):
        self._captions, self._split, self._every, self._inference = True, split, every, inference
        self._points, self._samples = {}, []
        self._feat_dir = None
        import os
        if root is not None and os.path.exists(os.path.join(root, "splits", split_id, split + ".txt")):
            # dataset.py:35-52: the points of this split from annotations/points.txt + captions.txt; frames as features
            from .dataset import TennisSet
            if feats_model is None:
                raise ValueError("the caption source on disk reads pre-extracted features: pass feats_model (dataset.py:42-44)")
            ts = TennisSet(root=root, split=split, split_id=split_id, every=1, balance=False, feats_model=feats_model)
            for pid, pt in ts._points.items():
                self._points[pid] = [pt[0], int(pt[1]), int(pt[2]), 0, pt[-1]]
            self._samples = list(self._points.keys())
            self._feat_dir, self._feat_path = ts.feat_dir, ts.get_feature_path
        else:
            rng = np.random.default_rng(zlib.crc32(f"{split}:{seed}".encode()))
            for i in range(n_points):
                start = int(rng.integers(0, 1000))
                n = int(np.clip(rng.normal(mean_frames, mean_frames / 3), 4, 3 * mean_frames))
                cap = " ".join(rng.choice(WORDS, size=int(rng.integers(4, 14))))
                pid = f"P{split}{i:04d}"
                self._points[pid] = ["V006", start, start + n, 0, cap]
                self._samples.append(pid)
        self._fdim, self._seed = feature_dim, seed
        if vocab is None:                                              # dataset.py:55-58
            counter = {}
            for p in self._points.values():
                for w in p[4].split():
                    counter[w] = counter.get(w, 0) + 1
            self.vocab = Vocab(counter)
        else:
            self.vocab = vocab
        for pid in self._samples:                                      # dataset.py:62-74
            toks = self._points[pid][4].split()
            ids = self.vocab[toks[:max_cap_len] if max_cap_len >= 0 else toks]
            ids = [self.vocab[self.vocab.bos_token]] + ids + [self.vocab[self.vocab.eos_token]]
            self._points[pid].append(np.array(ids, dtype=np.int32))

    def __len__(self):
        return len(self._samples)

    def get_captions(self, ids=False, split=False):                    # dataset.py:76-91
        caps = [self._points[s][5] if ids else self._points[s][4] for s in self._samples]
        return [c.split() for c in caps] if split and not ids else caps

    def _feature(self, vid, frame):
        if self._feat_dir is not None:                                  # dataset.py:169-171
            return np.load(self._feat_path(self._feat_dir, vid, frame)).astype(np.float32)
        s = zlib.crc32(f"{vid}:{frame}:{self._seed}".encode())
        return np.abs(np.random.default_rng(s).normal(0, 1, self._fdim)).astype(np.float32) * 0.5

    def __getitem__(self, idx):                                        # dataset.py:154-183
        point = self._points[self._samples[idx]]
        vid, start, end, cap = point[0], int(point[1]), int(point[2]), point[5]
        imgs = np.stack([self._feature(vid, f) for c, f in enumerate(range(start, end)) if c % self._every == 0])
        if self._inference:
            return imgs, cap, len(imgs), len(cap), idx
        return imgs, cap, len(imgs), len(cap)

    def get_data_lens(self):                                           # dataset.py:235-247 (off by one kept)
        return [(int((int(self._points[s][2]) - int(self._points[s][1]) + 1) / self._every), len(self._points[s][5]))
                for s in self._samples]


def pad_batchify(samples):
    """``btf.Tuple(Pad(), Pad(), Stack('float32'), Stack('float32')[, Stack()])`` (utils/captioning.py:33-37)."""
    tmax = max(s[0].shape[0] for s in samples)
    lmax = max(len(s[1]) for s in samples)
    src = np.zeros((len(samples), tmax, samples[0][0].shape[1]), np.float32)
    tgt = np.zeros((len(samples), lmax), np.int32)
    for i, s in enumerate(samples):
        src[i, :s[0].shape[0]] = s[0]
        tgt[i, :len(s[1])] = s[1]
    out = [src, tgt, np.array([s[2] for s in samples], np.float32), np.array([s[3] for s in samples], np.float32)]
    if len(samples[0]) > 4:
        out.append(np.array([s[4] for s in samples], np.int64))
    return tuple(out)


def bucketed_batches(dataset, batch_size, num_buckets=5, shuffle=False, seed=0, epoch=0, rank=0, world=1):
    """Stand-in for ``FixedBucketSampler(lengths, batch_size, num_buckets, shuffle)`` with constant-width buckets:
    samples are grouped by target length so padding stays small; instance ids travel with the batch, evaluate()
    restores the dataset order.  ``shuffle=False`` is the reference's validation / test sampler
    (utils/captioning.py:62-86); ``shuffle=True`` its TRAINING sampler (:48-55): the samples of a bucket are permuted
    before they are cut into batches and the batches are visited in random order, afresh every epoch
    (``default_rng(seed + epoch)``).  ``rank`` / ``world``: a data-parallel rank takes batches rank::world of that
    (identically seeded) list, padded by wrapping so that every rank runs the same number of steps."""
    lens = [l[-1] for l in dataset.get_data_lens()]
    lo, hi = min(lens), max(lens)
    width = max(1, math.ceil((hi - lo + 1) / num_buckets))
    buckets = {}
    for i, l in enumerate(lens):
        buckets.setdefault((l - lo) // width, []).append(i)
    rng = np.random.default_rng(seed + epoch) if shuffle else None
    batches = []
    for k in sorted(buckets):
        idxs = buckets[k]
        if shuffle:
            idxs = [idxs[j] for j in rng.permutation(len(idxs))]
        batches += [idxs[s:s + batch_size] for s in range(0, len(idxs), batch_size)]
    if shuffle:
        batches = [batches[j] for j in rng.permutation(len(batches))]
    if world > 1:
        n = -(-len(batches) // world) * world
        batches = [batches[j % len(batches)] for j in range(n)][rank::world]
    for ids in batches:
        yield pad_batchify([dataset[i] for i in ids])


def write_sentences(sentences, file_path):                             # utils/captioning.py:89-95
    with io.open(file_path, "w", encoding="utf-8") as of:
        for sent in sentences:
            of.write((u" ".join(sent) if isinstance(sent, (list, tuple)) else sent) + u"\n")


def evaluate(data_loader, model, translator, data_train):
    """reference train_gnmt.py:264-302: teacher-forced MaskedSoftmaxCELoss + beam search; returns
    (avg_loss, translations ordered by instance id).  Everything numeric runs on the GPU."""
    from .engine import masked_softmax_ce
    translation_out, all_inst_ids = [], []
    avg_loss_denom, avg_loss = 0, 0.0
    for src_seq, tgt_seq, src_valid_length, tgt_valid_length, inst_ids in data_loader:
        src = model.embed_source(torch.from_numpy(src_seq).cuda())     # frame mode: the clip's frames through the CNN
        tgt = torch.from_numpy(tgt_seq).cuda()
        svl = torch.from_numpy(src_valid_length).cuda()
        tvl = torch.from_numpy(tgt_valid_length).cuda()
        b, t = src.shape[0], src.shape[1]
        cap = model._captioner(translator._beam_size, translator._max_length, b, t)
        cap.encode(src, svl)
        out = cap.decode_seq(tgt[:, :-1])                              # model(src, tgt[:, :-1], ...)   :280
        loss = masked_softmax_ce(out, tgt[:, 1:], tvl - 1).mean().item()   # :281
        all_inst_ids.extend(inst_ids.astype(np.int32).tolist())
        avg_loss += loss * (tgt_seq.shape[1] - 1)
        avg_loss_denom += (tgt_seq.shape[1] - 1)
        samples, _, sample_valid_length = translator.translate(src, svl)   # :287-288
        max_score_sample = samples[:, 0, :].cpu().numpy()
        svl0 = sample_valid_length[:, 0].cpu().numpy()
        for i in range(max_score_sample.shape[0]):                     # :291-294
            translation_out.append([data_train.vocab.idx_to_token[ele]
                                    for ele in max_score_sample[i][1:(svl0[i] - 1)]])
    avg_loss = avg_loss / avg_loss_denom
    real_translation_out = [None for _ in range(len(all_inst_ids))]
    for ind, sentence in zip(all_inst_ids, translation_out):           # :298-300
        real_translation_out[ind] = sentence
    return avg_loss, real_translation_out
