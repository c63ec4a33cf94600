"""Host side of the temporal-head training path — counterparts of reference train.py:
``gluon.Trainer(params, 'sgd', {...})`` (:298-299), ``gluon.loss.SoftmaxCrossEntropyLoss`` (:324) and
``train_model`` (:388-499) for ``CNNRNN(model=None, type='gru'|'lstm')`` on pre-extracted features (the frozen-backbone
recipe, e.g. model 0042 of models/README.md:57-59; ``engine.TemporalHeadTrainer``) and for the end-to-end frame classifier
``FrameModel(DenseNet121.features, classes)`` with BatchNorm in training mode (model 0006; ``engine.FrameModelTrainer``) — both
expose ``forward_backward(x, labels) -> (loss, logits)``, ``step(batch_size, lr, momentum, wd)``, ``grads`` and ``state_dict()``.
All arithmetic runs in libtennis_hip (tn_head_* / tn_finetune_*); torch is the
buffer / collective plumbing: with ``torch.distributed`` initialised (backend nccl = RCCL) every rank trains on its
shard of the batch and the flat gradient buffer is all-reduced before the update.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .engine import TemporalHeadTrainer


def allreduce_and_step(trainer, global_batch_size: int, lr: float, momentum: float, wd: float):
    """``trainer.step(batch_size)`` of a data-parallel run: sum the per-rank gradients of the summed losses, then
    Gluon's rescale 1/batch_size with the GLOBAL batch size (reference: one Trainer over all devices,
    ``trainer.step(FLAGS.batch_size)``, train.py:424)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(trainer.grads, op=dist.ReduceOp.SUM)
    trainer.step(global_batch_size, lr, momentum, wd)


class Trainer:
    """``gluon.Trainer(model.collect_params(), 'sgd', {'learning_rate', 'momentum', 'wd'})`` for the temporal head."""

    def __init__(self, head: TemporalHeadTrainer, optimizer: str = "sgd", optimizer_params: dict | None = None):
        if optimizer != "sgd":
            raise ValueError("only 'sgd' is built (reference train.py:298)")
        op = dict(optimizer_params or {})
        self.head = head
        self.learning_rate = float(op.get("learning_rate", 0.01))
        self.momentum = float(op.get("momentum", 0.0))
        self.wd = float(op.get("wd", 0.0))

    def set_learning_rate(self, lr: float):
        self.learning_rate = float(lr)

    def step(self, batch_size: int):
        allreduce_and_step(self.head, batch_size, self.learning_rate, self.momentum, self.wd)


def train_model(head: TemporalHeadTrainer, train_batches, metrics, trainer: Trainer, epochs: int, batch_size: int,
                lr_steps=(10, 20), lr_factor: float = 0.75, start_epoch: int = 0, val_fn=None, save_dir: str | None = None,
                log=print):
    """reference train.py:388-499 with the model call, loss and backward fused into ``head.forward_backward``.
    ``train_batches``: callable -> iterable of (features (B,T,F) tensor, labels (B,) tensor) per epoch."""
    lr_counter = 0
    lr_steps = list(lr_steps) + [1 << 30]
    history = []
    for epoch in range(start_epoch, epochs):
        if epoch == lr_steps[lr_counter]:                                   # :395-397
            trainer.set_learning_rate(trainer.learning_rate * lr_factor)
            lr_counter += 1
        for m in metrics:
            m.reset()
        train_sum_loss, nb = 0.0, 0
        for x, y in train_batches():
            loss, logits = head.forward_backward(x, y)                      # :415-421
            trainer.step(batch_size)                                        # :424
            train_sum_loss += float(loss.mean())                            # :427
            nb += 1
            for m in metrics:
                m.update([y], [logits])                                     # :430-431
        row = {"epoch": epoch, "lr": trainer.learning_rate, "loss": train_sum_loss / max(1, nb)}
        if val_fn is not None:
            row["val"] = val_fn(head)
        history.append(row)
        log("[Epoch {}] loss: {:.3f} lr: {:.2E}".format(epoch, row["loss"], trainer.learning_rate))
        if save_dir:                                                        # :497
            os.makedirs(save_dir, exist_ok=True)
            with open(os.path.join(save_dir, "{:04d}.params".format(epoch)), "wb") as f:
                np.savez(f, **head.state_dict())
    return history
