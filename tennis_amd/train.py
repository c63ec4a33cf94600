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
        comm = _grad_comm(trainer.grads.device)
        if comm is not None:
            comm.allreduce_(trainer.grads).wait()      # tn_allreduce_f32: RCCL behind the C-ABI
        else:
            dist.all_reduce(trainer.grads, op=dist.ReduceOp.SUM)
    trainer.step(global_batch_size, lr, momentum, wd)


_COMMS = {}


def _grad_comm(device):
    """The process's C-ABI communicator over the default group (GPU runs; None on CPU: the gloo tests)."""
    if device.type != "cuda":
        return None
    if device.index not in _COMMS:
        from . import sharding
        _COMMS[device.index] = sharding.feature_comm(device)
    return _COMMS[device.index]


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


def best_epoch_from_scores(scores_path: str):
    """(epoch, score) of the best line of ``scores.txt`` ("epoch<TAB>AVG_NB_f1" per validated epoch, written by
    ``train_model``; read back the way reference evaluate.py:186-195 / train.py:334-342 do: first strict maximum)."""
    best_epoch, best_score = -1, -1.0
    with open(scores_path, "r") as f:
        for line in f:
            parts = line.rstrip().split()
            if len(parts) != 2:
                continue
            if float(parts[1]) > best_score:
                best_epoch, best_score = int(parts[0]), float(parts[1])
    return best_epoch, best_score


def newest_params(save_dir: str):
    """Newest ``NNNN.params`` of an experiment directory or None (reference train.py:287-293, evaluate.py:206-212)."""
    if not os.path.isdir(save_dir):
        return None
    files = sorted((f for f in os.listdir(save_dir) if f.endswith(".params")), reverse=True)
    return os.path.join(save_dir, files[0]) if files else None


def is_main_rank() -> bool:
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def train_model(head: TemporalHeadTrainer, train_batches, metrics, trainer: Trainer, epochs: int, batch_size: int,
                lr_steps=(10, 20), lr_factor: float = 0.75, start_epoch: int = 0, val_fn=None, save_dir: str | None = None,
                log=print, model=None, score_key: str = "AVG_NB_f1"):
    """reference train.py:388-499 with the model call, loss and backward fused into ``head.forward_backward``.
    ``train_batches``: callable -> iterable of (features (B,T,F) tensor, labels (B,) tensor) per epoch.
    Per epoch (rank 0 only): the validation ``AVG_NB_f1`` is appended to ``<save_dir>/scores.txt`` (:487-489) and the
    parameters go to ``<save_dir>/NNNN.params`` (:497) — through ``model.save_parameters`` (the MXNet container with
    Gluon's structural names, what the reference's ``load_parameters`` reads) when the mirror ``model`` is given,
    else an ``.npz`` of the trainer's prefixed names under ``NNNN.npz``."""
    lr_counter = 0
    lr_steps = list(lr_steps) + [1 << 30]
    history = []
    for epoch in range(0, start_epoch):                                     # a resumed run keeps the schedule it had
        if epoch == lr_steps[lr_counter]:
            trainer.set_learning_rate(trainer.learning_rate * lr_factor)
            lr_counter += 1
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
        if save_dir and is_main_rank():
            os.makedirs(save_dir, exist_ok=True)
            if isinstance(row.get("val"), dict) and score_key in row["val"]:   # :487-489
                with open(os.path.join(save_dir, "scores.txt"), "a") as f:
                    f.write(str(epoch) + "\t" + str(float(row["val"][score_key])) + "\n")
            if model is not None:                                           # :497
                model.set_params(head.state_dict())
                model.save_parameters(os.path.join(save_dir, "{:04d}.params".format(epoch)))
            else:
                with open(os.path.join(save_dir, "{:04d}.npz".format(epoch)), "wb") as f:
                    np.savez(f, **head.state_dict())
    return history


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="tennis_amd train (flags of reference train.py:30-95)")
    p.add_argument("--backbone", default="DenseNet121")
    p.add_argument("--freeze_backbone", action="store_true")
    p.add_argument("--no_augment", action="store_true", help="test transform for the train split too (not a reference flag)")
    p.add_argument("--model_id", default="0000")
    p.add_argument("--split_id", default="02")
    p.add_argument("--data_shape", type=int, default=224)
    p.add_argument("--every", default="1, 1, 1")
    p.add_argument("--balance", default="True, False, False")
    p.add_argument("--window", type=int, default=1)
    p.add_argument("--padding", type=int, default=1)
    p.add_argument("--stride", type=int, default=1)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--epochs", type=int, default=20)
    p.add_argument("--lr", type=float, default=0.001)
    p.add_argument("--lr_factor", type=float, default=0.75)
    p.add_argument("--lr_steps", default="10, 20")
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--wd", type=float, default=0.0001)
    p.add_argument("--feats_model", default=None)
    # the rest of the reference's flags (train.py:32-93), accepted so that its command lines parse unchanged
    p.add_argument("--backbone_from_id", default=None, help="start the frame model from the newest .params of this model id (train.py:222-235)")
    p.add_argument("--log_interval", type=int, default=100)
    p.add_argument("--num_gpus", type=int, default=1, help="one process per GPU here: data parallelism comes from torch.distributed (train.allreduce_and_step)")
    p.add_argument("--vis", action="store_true", help="refused: outside the accelerated path (SURVEY 2a)")
    p.add_argument("--flow", default="", help="anything but '' is refused: outside the accelerated path (SURVEY 2a)")
    p.add_argument("--max_batches", type=int, default=-1, help="stop an epoch after this many batches (train.py:92: 'for 0031')")
    p.add_argument("--temp_pool", default=None, help="gru or lstm (trained); mean / max need no training")
    p.add_argument("--root", default="data")
    p.add_argument("--decode", default="device", choices=["device", "host", "auto"], help="where on-disk JPEG frames are decoded (see evaluate.py)")
    p.add_argument("--num_workers", type=int, default=3, help="loader threads (train.py:101-102), see evaluate.py")
    p.add_argument("--frames_per_video", type=int, default=16)
    p.add_argument("--exp_root", default=os.path.join("models", "vision", "experiments"))
    return p


def main(argv=None):
    """reference train.py::main (:98-386) for the two trainable configurations on the hot path:
      * ``--feats_model <id> --window W --temp_pool gru|lstm``: the temporal head on pre-extracted features (backbone frozen);
      * ``--window 1`` without ``--feats_model``: the frame classifier end to end (BatchNorm in training mode).
    End-to-end training on frames from disk uses the reference's TRAIN transform for the train split (RandomResizedCrop,
    RandomFlipLeftRight, RandomColorJitter(0.4, 0.4, 0.4), RandomLighting(0.1); train.py:125-139 - round 4: one GPU launch group per
    batch, ``tennis_amd.transforms``), the test transform for validation; ``--no_augment`` keeps the test transform everywhere."""
    from . import transforms
    from .dataset import DataLoader, TennisSet
    from .engine import FrameModelTrainer, TemporalHeadTrainer
    from .sharding import init_distributed
    from .evaluate import evaluate_model
    from .metrics.vision import PRF1
    from .model_zoo import get_model
    from .models.vision.definitions import CNNRNN, FrameModel
    from . import weights as W
    flags = build_parser().parse_args(argv)
    if flags.flow or flags.vis:
        raise NotImplementedError("--flow / --vis: optical-flow input and visualisation are outside the accelerated path (SURVEY 2a)")
    if flags.num_workers < 0:                          # the reference's -1 = cpu_count() (train.py:101-102)
        flags.num_workers = 3                                               # (files -> features peaks at three decoder threads: scripts/bench_pipeline.py, profiles/r05_c_*)
    every = [int(s) for s in flags.every.split(",")]
    balance = [s.strip().lower() in ("true", "t") for s in flags.balance.split(",")]
    lr_steps = [int(s) for s in flags.lr_steps.split(",")]
    rank, world, dev = init_distributed()              # one process per GPU under torchrun, (0, 1, cuda:current) otherwise
    if flags.batch_size % world:
        raise SystemExit(f"--batch_size {flags.batch_size} must be a multiple of the {world} ranks")
    local_bs = flags.batch_size // world
    # frames on disk come at their native size: Resize(s + 32) / CenterCrop(s) / ToTensor / Normalize as evaluate.py:93-98
    # (one GPU launch per batch); the synthetic source already produces data_shape frames
    on_disk = flags.feats_model is None and os.path.exists(os.path.join(flags.root, "splits", flags.split_id, "train.txt"))
    tf = transforms.Compose([transforms.Resize(flags.data_shape + 32), transforms.CenterCrop(flags.data_shape),
                             transforms.ToTensor(),
                             transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])]) if on_disk else None
    # train.py:125-139: jitter_param = 0.4, lighting_param = 0.1 (every rank draws from its own seed: the ranks train on different rows)
    tf_train = transforms.Compose([transforms.RandomResizedCrop(flags.data_shape), transforms.RandomFlipLeftRight(),
                                   transforms.RandomColorJitter(brightness=0.4, contrast=0.4, saturation=0.4),
                                   transforms.RandomLighting(0.1), transforms.ToTensor(),
                                   transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])],
                                  seed=1000 + rank) if on_disk and not flags.no_augment else tf
    mk = lambda split, ev, bal: TennisSet(root=flags.root, transform=tf_train if split == "train" else tf, split=split, every=ev, padding=flags.padding,
                                          stride=flags.stride, window=flags.window, model_id=flags.model_id,
                                          split_id=flags.split_id, balance=bal, feats_model=flags.feats_model,
                                          data_shape=flags.data_shape, frames_per_video=flags.frames_per_video,
                                          decode=flags.decode)
    train_set, val_set = mk("train", every[0], balance[0]), mk("val", every[1], balance[1])
    n_cls = len(train_set.classes)
    save_dir = os.path.join(flags.exp_root, flags.model_id)
    if flags.feats_model is not None:
        if flags.window <= 1 or flags.temp_pool not in ("gru", "lstm"):
            raise SystemExit("training on features needs --window > 1 and --temp_pool gru|lstm (definitions.py:94-96)")
        feat_dim = int(np.asarray(train_set[0][0]).shape[-1])
        model = CNNRNN(None, num_classes=n_cls, type=flags.temp_pool, hidden_size=128)
        model.rnn._materialize(feat_dim)
        model.classes._materialize(256)
        last = "keep"
        mk_head = lambda p: TemporalHeadTrainer(p, feat_dim, 128, n_cls, max_batch=local_bs, max_steps=flags.window,
                                                rnn_prefix=model.rnn.prefix, dense_prefix=model.classes.prefix,
                                                type=flags.temp_pool)
    else:
        if flags.window != 1 or flags.freeze_backbone:
            raise SystemExit("end-to-end training is built for --window 1 with a trainable backbone; a frozen backbone "
                             "trains on features (--feats_model after evaluate --save_feats)")
        model = FrameModel(get_model(flags.backbone, pretrained=True).features, n_cls)
        model.initialize()
        model.classes._materialize(1024)
        if flags.backbone_from_id:                     # train.py:222-235 loads it for window > 1 only; a frame model started from
            bb = newest_params(os.path.join(flags.exp_root, flags.backbone_from_id))   # another experiment is the same request
            if bb is not None:
                model.load_parameters(bb)
                print("Loaded backbone params: {}".format(bb))
        last = "discard"
        mk_head = lambda p: FrameModelTrainer(p, flags.data_shape, n_cls, batch=local_bs, prefix=model.backbone.prefix,
                                              dense_prefix=model.classes.prefix)
    # resume from the newest NNNN.params of the experiment (train.py:286-295)
    start_epoch = 0
    newest = newest_params(save_dir)
    if newest is not None:
        model.load_parameters(newest)
        start_epoch = int(os.path.basename(newest).split(".")[0]) + 1
        print("Loaded model params: {}".format(newest))
    head = mk_head({k: v.data for k, v in model.collect_params().items()})
    train_data = DataLoader(train_set, flags.batch_size, shuffle=True, last_batch=last, num_workers=flags.num_workers)
    val_data = DataLoader(val_set, flags.batch_size, shuffle=False, num_workers=flags.num_workers)
    trainer = Trainer(head, "sgd", {"learning_rate": flags.lr, "momentum": flags.momentum, "wd": flags.wd})
    metrics = [PRF1(label_names=train_set.classes)]

    def batches():          # every rank walks the same (seeded) batches and trains on its rows rank::world of each
        for i, (data, labels, _) in enumerate(train_data):
            if flags.max_batches > 0 and i > flags.max_batches:             # train.py:405
                break
            x = data if isinstance(data, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(data))
            y = torch.from_numpy(labels.astype(np.int32))
            if world > 1:
                if x.shape[0] % world:
                    continue        # a ragged last batch cannot be split evenly: dropped in data-parallel runs
                x, y = x[rank::world], y[rank::world]
            yield x.to(dev), y.to(dev)

    def validate(h):                                     # train.py:445-470: the validation metrics of the updated model
        model.set_params(h.state_dict())
        vm = [PRF1(label_names=val_set.classes)]
        evaluate_model(model, val_data, val_set, vm)
        return dict(vm[0].get())

    hist = train_model(head, batches, metrics, trainer, flags.epochs, flags.batch_size, lr_steps=lr_steps,
                       lr_factor=flags.lr_factor, start_epoch=start_epoch, val_fn=validate, save_dir=save_dir, model=model)
    if hist and is_main_rank():
        best = max(hist, key=lambda r: r["val"].get("AVG_NB_f1", 0.0))
        print("[Finished] best epoch {} val AVG_NB_f1={:.3f}".format(best["epoch"], best["val"].get("AVG_NB_f1", 0.0)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
