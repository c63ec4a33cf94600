"""ORACLE (test infrastructure only) — CPU restatement of the reference's vision
model wrappers and metric, line by line from the in-repo Python:

  FrameModel          reference models/vision/definitions.py:10-33
  TimeDistributed     reference utils/layers.py:38-46 (default 'reshape' style)
  TemporalPooling     reference models/vision/definitions.py:36-72
  CNNRNN              reference models/vision/definitions.py:75-110
  PRF1                reference metrics/vision.py:27-99

The wrappers' own logic is pinned by the reference source above; the ops they
call (backbone, rnn, Dense) are the [EXT] restatements in densenet_np / rnn_np
(PARITY UNPINNED for those, see their headers).  PRF1 is pinned against golden
vectors produced by the reference's own class (tests/golden/prf1_*.json, made
by tests/golden/make_reference_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import numpy as np

from . import densenet_np as dn
from . import rnn_np as rn


def frame_model(x, p, num_classes=11, backbone_prefix="densenet0_", cls_prefix="framemodel0_dense0_"):
    """definitions.py:27-33 (swap=False): logits = classes(backbone(x)) if classes else feats."""
    f = dn.densenet121_features(x, p, backbone_prefix)
    return dn.dense(f, p, cls_prefix) if num_classes > 0 else f


def time_distributed(fn, x):
    """layers.py:39-46: merge dims 0,1 -> apply -> split dim 0 back to (B,T)."""
    b, t = x.shape[:2]
    y = fn(x.reshape((b * t,) + x.shape[2:]))
    if isinstance(y, tuple):
        return tuple(yi.reshape((b, t) + yi.shape[1:]) for yi in y)
    if isinstance(y, list):
        return [yi.reshape((b, t) + yi.shape[1:]) for yi in y]
    return y.reshape((b, t) + y.shape[1:])


def temporal_pooling(x, p, pool="max", feats=False, has_classes=True,
                     backbone_prefix="densenet0_", cls_prefix="framemodel0_dense0_"):
    """definitions.py:63-72: [td] -> mean|max over axis 1 -> classes."""
    if not feats:
        x = time_distributed(lambda z: dn.densenet121_features(z, p, backbone_prefix), x)
    x = x.mean(axis=1, dtype=np.float32) if pool == "mean" else x.max(axis=1)
    return dn.dense(x, p, cls_prefix) if has_classes else x


def cnnrnn(x, p, mode="gru", feats=True, num_classes=11, backbone_prefix="densenet0_",
           rnn_prefix="cnnrnn0_gru0_", cls_prefix="cnnrnn0_dense0_"):
    """definitions.py:103-110: [td] -> bi-rnn(NTC) -> max over T -> classes."""
    if not feats:
        x = time_distributed(lambda z: dn.densenet121_features(z, p, backbone_prefix), x)
    seq, _, _ = rn.birnn_layer(x, p, rnn_prefix, mode)
    pooled = seq.max(axis=1)
    out = dn.dense(pooled, p, cls_prefix) if num_classes > 0 else pooled
    return out, seq


class PRF1:
    """metrics/vision.py:8-99 without the mx.metric.EvalMetric base.

    Quirk kept: '*_prec' = TP/#label and '*_rec' = TP/#pred.
    """

    def __init__(self, label_names):
        self.label_names = list(label_names)
        self.reset()

    def reset(self):  # vision.py:94-99
        n = len(self.label_names)
        self.scores = np.zeros((3, n))
        self.mat = np.zeros((n, n))

    def update(self, labels, preds):  # vision.py:27-58; lists of arrays
        for label, pred in zip(labels, preds):
            pred = np.asarray(pred)
            label = np.asarray(label)
            if pred.shape != label.shape:
                pred = pred.argmax(axis=1)
            pred = pred.astype("int32")
            label = label.astype("int32")
            for i in range(len(label)):
                self.mat[label[i], pred[i]] += 1
            for i in range(len(self.label_names)):
                predictions = pred == i
                positives = label == i
                self.scores[0, i] += np.logical_and(predictions, positives).sum()
                self.scores[1, i] += positives.sum()
                self.scores[2, i] += predictions.sum()

    def get(self):  # vision.py:60-92
        eps = np.finfo(float).eps
        scores, ap, ar, af = [], [], [], []
        for i, c in enumerate(self.label_names):
            prec = self.scores[0][i] / (self.scores[1][i] + eps)
            rec = self.scores[0][i] / (self.scores[2][i] + eps)
            f1 = 2 * (prec * rec) / (prec + rec + eps)
            scores += [(c + "_prec", prec), (c + "_rec", rec), (c + "_f1", f1)]
            ap.append(prec); ar.append(rec); af.append(f1)
        scores.append(("AVG_prec", sum(ap) / len(ap)))
        scores.append(("AVG_rec", sum(ar) / len(ar)))
        scores.append(("AVG_f1", sum(af) / len(af)))
        scores.append(("AVG_NB_prec", sum(ap[1:]) / len(ap[1:])))
        scores.append(("AVG_NB_rec", sum(ar[1:]) / len(ar[1:])))
        scores.append(("AVG_NB_f1", sum(af[1:]) / len(af[1:])))
        return scores
