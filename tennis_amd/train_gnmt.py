"""Host side of the captioner's training path — counterpart of reference train_gnmt.py::train (:305-470) for the
configuration it ships as default (``--num_layers 2 --num_bi_layers 1``, ``--cell_type gru`` or ``lstm``):

    trainer = gluon.Trainer(model.collect_params(), 'adam', {'learning_rate': lr})                      :310
    for epoch: for batch in train loader (FixedBucketSampler over target lengths):                      :318
        out, _ = model(src, tgt[:, :-1], src_valid_length, tgt_valid_length - 1)                        :331
        loss = loss_function(out, tgt[:, 1:], tgt_valid_length - 1).mean() * (L - 1) / mean(valid - 1)  :332-333
        loss.backward(); trainer.step(1)                                                                 :334,337
      evaluate valid / test: loss, BLEU of the beam-search translations, write them out                 :372-447
      keep the parameters with the best validation BLEU; lr *= lr_update_factor once
      epoch + 1 >= 2/3 of the epochs; save the epoch's parameters                                       :450-461

All arithmetic runs in libtennis_hip (``tn_gnmt_trainer_*`` for the step, ``tn_gnmt_*`` for evaluation); with
``torch.distributed`` initialised (backend nccl = RCCL) every rank trains on its own batches and the flat gradient
buffer is averaged over ranks before the Adam update.  nlg-eval's METEOR / CIDEr stay external, as in the survey.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .captions import bucketed_batches, evaluate, write_sentences
from .engine import GNMTTrainer
from .metrics.bleu import compute_bleu


def allreduce_grads(trainer: GNMTTrainer):
    """Data parallelism: every rank's loss is already a per-token average, so the ranks' gradients are averaged."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        g = trainer.grads
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        g /= dist.get_world_size()


def train(data_train, data_val, data_test, model, translator, epochs: int, batch_size: int, lr: float = 1e-3,
          lr_update_factor: float = 0.5, dropout: float = 0.0, num_buckets: int = 5, test_batch_size: int = 32,
          start_epoch: int = 0, save_dir: str | None = None, seed: int = 0, log=print):
    """-> history: one dict per epoch (train loss, valid / test loss and BLEU, learning rate)."""
    enc = model.encoder
    if enc._cell_type not in ("gru", "lstm") or enc._num_layers != 2 or enc._num_bi_layers != 1:
        raise NotImplementedError("the training step is built for num_layers=2, num_bi_layers=1 (the reference's flag defaults)")
    params = {k: v.data for k, v in model.collect_params().items()}
    max_t = max(l[0] for l in data_train.get_data_lens())
    max_l = max(l[-1] for l in data_train.get_data_lens())
    trainer = GNMTTrainer(params, model._input_size, enc._hidden_size, model._embed_size, len(model.tgt_vocab),
                          max_batch=batch_size, max_src_len=max_t, max_tgt_len=max_l, prefix=model.prefix,
                          cell_type=enc._cell_type)
    if dropout > 0:
        trainer.set_dropout(dropout, seed)
    val_tgt = data_val.get_captions(split=True) if data_val is not None else None
    test_tgt = data_test.get_captions(split=True) if data_test is not None else None
    best_valid_bleu, history = 0.0, []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    for epoch_id in range(start_epoch, epochs):
        tot, nb = 0.0, 0
        for src, tgt, svl, tvl, *_ in bucketed_batches(data_train, batch_size, num_buckets):
            loss = trainer.forward_backward(torch.from_numpy(src).cuda(), torch.from_numpy(svl.astype(np.int32)).cuda(),
                                            torch.from_numpy(tgt).cuda(), torch.from_numpy(tvl.astype(np.int32)).cuda())
            allreduce_grads(trainer)
            trainer.step(lr)                                                     # trainer.step(1)
            tot += float(loss)
            nb += 1
        rec = {"epoch": epoch_id, "train_loss": tot / max(1, nb), "lr": lr}
        model.set_params(trainer.state_dict())                                   # evaluation runs on the updated weights
        for name, ds, ref in (("valid", data_val, val_tgt), ("test", data_test, test_tgt)):
            if ds is None:
                continue
            ev_loss, out = evaluate(bucketed_batches(ds, test_batch_size, num_buckets), model, translator, data_train)
            bleu = compute_bleu([ref], out)[0]
            rec[f"{name}_loss"], rec[f"{name}_bleu"] = ev_loss, bleu
            log("[Epoch {}] {} Loss={:.4f}, {} ppl={:.4f}, {} bleu={:.2f}".format(epoch_id, name, ev_loss, name,
                                                                                 math.exp(min(ev_loss, 50.0)), name, bleu * 100))
            if save_dir:
                write_sentences(out, os.path.join(save_dir, "epoch{:d}_{}_out.txt".format(epoch_id, name)))
        if save_dir and rec.get("valid_bleu", 0.0) > best_valid_bleu:            # :450-454
            best_valid_bleu = rec["valid_bleu"]
            model.save_parameters(os.path.join(save_dir, "valid_best.params"), structural=False)
        if epoch_id + 1 >= (epochs * 2) // 3:                                    # :456-459
            lr *= lr_update_factor
            log("Learning rate change to {}".format(lr))
        if save_dir:
            model.save_parameters(os.path.join(save_dir, "{:04d}.params".format(epoch_id)), structural=False)
        history.append(rec)
    return history
