"""Generates golden vectors by importing the REFERENCE's own pure-Python pieces in
the build container (they cannot travel to the GPU box; the vectors can):

  * metrics/vision.py::PRF1  — imported with a ~20-line stub `mxnet` module
    (only EvalMetric / check_label_shapes / ndarray.argmax are touched);
  * metrics/bleu.py::compute_bleu — imports unmodified (stdlib + six).

Run:  python tests/golden/make_reference_golden.py   (needs /root/reference)
Writes tests/golden/prf1_reference.json and tests/golden/bleu_reference.json.
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub_mxnet():
    mx = types.ModuleType("mxnet")
    metric = types.ModuleType("mxnet.metric")

    class EvalMetric:
        def __init__(self, name, output_names=None, label_names=None, **kwargs):
            self.name = name

    def check_label_shapes(labels, preds, wrap=False, shape=False):
        if wrap:
            if not isinstance(labels, list):
                labels = [labels]
            if not isinstance(preds, list):
                preds = [preds]
        return labels, preds

    metric.EvalMetric = EvalMetric
    metric.check_label_shapes = check_label_shapes

    class _ND:  # minimal NDArray: shape, asnumpy
        def __init__(self, a):
            self.a = np.asarray(a)
            self.shape = self.a.shape

        def asnumpy(self):
            return self.a

    nd = types.ModuleType("mxnet.ndarray")
    nd.argmax = lambda x, axis: _ND(x.a.argmax(axis=axis).astype(np.float32))
    mx.metric, mx.ndarray, mx.nd = metric, nd, nd
    mx._ND = _ND
    sys.modules.update({"mxnet": mx, "mxnet.metric": metric, "mxnet.ndarray": nd})
    return mx


def main():
    mx = _stub_mxnet()
    sys.path.insert(0, REF)
    from metrics.vision import PRF1  # noqa: E402  (reference class)
    classes = [l.strip() for l in open(os.path.join(REF, "data", "classes.names"))]
    cases = []
    for seed, n, batches in [(1, 256, 1), (7, 300, 3), (11, 40, 2)]:
        rng = np.random.RandomState(seed)
        m = PRF1(label_names=classes)
        for _ in range(batches):
            logits = rng.randn(n, len(classes)).astype(np.float32)
            labels = rng.randint(0, len(classes), n).astype(np.float32)
            # bias predictions towards the label so that matches are non-trivial
            logits[np.arange(n), labels.astype(int)] += 1.5
            m.update([mx._ND(labels)], [mx._ND(logits)])
        # inputs are regenerated in the test from (seed, n, batches) with the same RandomState calls
        cases.append(dict(seed=seed, n=n, batches=batches, scores=[[k, float(v)] for k, v in m.get()],
                          mat=m.mat.tolist()))
    with open(os.path.join(HERE, "prf1_reference.json"), "w") as f:
        json.dump(dict(classes=classes, cases=cases), f)

    from metrics.bleu import compute_bleu  # noqa: E402
    # reference_corpus_list[set][translation] -> tokens (two reference sets)
    refs = [[["the", "player", "serves", "the", "ball", "in"], ["a", "forehand", "return", "far", "right"],
             ["near", "player", "hits", "a", "backhand", "to", "the", "left"]],
            [["the", "near", "player", "serves", "in"], ["forehand", "return", "to", "the", "far", "right"],
             ["a", "backhand", "left"]]]
    hyps = [["the", "player", "serves", "in"], ["a", "forehand", "return", "far", "right"],
            ["near", "player", "hits", "backhand", "left"]]
    out = []
    for kwargs in [dict(), dict(smooth=True), dict(bpe=False, split_compound_word=False, lower_case=True),
                   dict(max_n=2)]:
        try:
            r = compute_bleu(refs, hyps, **kwargs)
            out.append(dict(kwargs=kwargs, result=[float(x) if not isinstance(x, (list, tuple)) else
                                                   [float(y) for y in x] for x in r]))
        except TypeError:
            pass
    # plain-text inputs (tokenized=False): the mteval-13a / v14-international tokenizers, BPE joins and
    # compound splitting, upper/lower case, an empty hypothesis
    trefs = [["The near player serves an ace, 120 mph (wide)!", "A fore-hand return; it's OUT.",
              "Far player's back@@ hand lob - long 3.5 m", "point won"],
             ["Near player serves a 120mph ace wide.", "The forehand return is out", "A back@@ hand lob goes long, 3.5m",
              "the point is won"]]
    thyps = ["The near player serves an ace (wide), 120 mph!", "A fore-hand return is out.",
             "far player's back@@ hand lob long - 3.5 m", ""]
    text = []
    for kwargs in [dict(tokenized=False), dict(tokenized=False, tokenizer="intl"), dict(tokenized=False, tokenizer=None),
                   dict(tokenized=False, lower_case=True, smooth=True),
                   dict(tokenized=False, bpe=True, split_compound_word=True, lower_case=True)]:
        r = compute_bleu(trefs, thyps, **kwargs)
        text.append(dict(kwargs=kwargs, result=[float(x) if not isinstance(x, (list, tuple)) else
                                                [float(y) for y in x] for x in r]))
    with open(os.path.join(HERE, "bleu_reference.json"), "w") as f:
        json.dump(dict(refs=refs, hyps=hyps, cases=out, text_refs=trefs, text_hyps=thyps, text_cases=text), f)
    print("wrote", len(cases), "PRF1 cases,", len(out), "BLEU cases")


if __name__ == "__main__":
    main()
