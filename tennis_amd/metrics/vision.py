"""Precision / Recall / F1 metric — mirror of reference metrics/vision.py::PRF1.

Same constructor, ``update/get/reset``, ``scores`` and ``mat`` attributes and the
same 39 ``(name, value)`` pairs, including the reference's swapped names
('*_prec' = TP/#label, '*_rec' = TP/#pred; vision.py:73-74).  The argmax and the
confusion histogram (vision.py:41-49, a per-sample Python loop in the reference)
run on the GPU for device tensors; the per-class counts follow from the matrix.
"""
import numpy as np
import torch

from .. import _lib


class PRF1:
    def __init__(self, axis=1, name="prf1", output_names=None, label_names=None):
        assert label_names is not None, "label_names cant be None"   # vision.py:21
        self.name = name
        self.axis = axis
        self.label_names = list(label_names)
        self.reset()

    def reset(self):                                                   # vision.py:94-99
        n = len(self.label_names)
        self.scores = np.zeros((3, n))
        self.mat = np.zeros((n, n))

    def update(self, labels, preds):                                   # vision.py:27-58
        if not isinstance(labels, (list, tuple)):
            labels, preds = [labels], [preds]
        n = len(self.label_names)
        for label, pred in zip(labels, preds):
            if isinstance(pred, torch.Tensor) and pred.is_cuda and pred.dim() == 2:
                ctx = _lib.default_context(pred.device.index)
                lab = (label if isinstance(label, torch.Tensor) else torch.as_tensor(np.asarray(label)))
                lab = lab.to(device=pred.device, dtype=torch.int32).contiguous()
                pr = pred.contiguous().float()
                mat = torch.zeros((n, n), dtype=torch.int64, device=pred.device)
                _lib.check(ctx.lib.tn_prf1_update(ctx.handle, _lib.ptr(pr), _lib.ptr(lab), pr.shape[0], n,
                                                  _lib.ptr(mat)), "tn_prf1_update")
                m = mat.cpu().numpy().astype(np.float64)
            else:  # host arrays of class indices or scores: same arithmetic in numpy
                p = pred.cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
                l = label.cpu().numpy() if isinstance(label, torch.Tensor) else np.asarray(label)
                if p.shape != l.shape:
                    p = p.argmax(axis=self.axis)
                p, l = p.astype("int32"), l.astype("int32")
                m = np.zeros((n, n))
                np.add.at(m, (l, p), 1)
            self.mat += m
            self.scores[0] += np.diag(m)          # matches      (vision.py:56)
            self.scores[1] += m.sum(axis=1)       # positives    (vision.py:57)
            self.scores[2] += m.sum(axis=0)       # predictions  (vision.py:58)

    def get(self):                                                     # vision.py:60-92
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
