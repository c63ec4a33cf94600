"""Captioner evaluation driver - counterpart of reference evaluate_gnmt.py: load the best parameters of a model id
(``valid_best.params``, else the newest epoch file) and report teacher-forced loss, perplexity and BLEU of the beam-search
translations on the validation and test splits, writing the sentences out (evaluate_gnmt.py:196-258; same flag names as
train_gnmt).  Feature mode (``--feats_model`` in the reference): the frame features are the inputs."""
from __future__ import annotations

import math
import os

from .captions import bucketed_batches, evaluate, write_sentences
from .metrics.bleu import compute_bleu
from .train_gnmt import build, build_parser


def main(argv=None):
    flags = build_parser().parse_args(argv)
    data_train, data_val, data_test, model, translator = build(flags)
    exp = os.path.join(flags.root, flags.model_id)
    path = os.path.join(exp, "valid_best.params")
    if not os.path.exists(path):
        files = sorted(f for f in os.listdir(exp) if f.endswith(".params")) if os.path.isdir(exp) else []
        if not files:
            raise FileNotFoundError(f"no parameter file under {exp}")
        path = os.path.join(exp, files[-1])
    model.load_parameters(path)
    print("Loaded params: {}".format(path))
    out = {}
    for name, ds in (("valid", data_val), ("test", data_test)):
        loss, sents = evaluate(bucketed_batches(ds, flags.test_batch_size, flags.num_buckets), model, translator, data_train)
        bleu = compute_bleu([ds.get_captions(split=True)], sents)[0]
        print("Best model {} Loss={:.4f}, {} ppl={:.4f}, {} bleu={:.2f}".format(name, loss, name, math.exp(min(loss, 50.0)), name, bleu * 100))
        write_sentences(sents, os.path.join(exp, "best_{}_out.txt".format(name)))
        out[name] = (loss, bleu)
    return out


if __name__ == "__main__":
    main()
