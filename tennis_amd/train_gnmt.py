"""Host side of the captioner's training path — counterpart of reference train_gnmt.py::train (:305-470): ``--cell_type gru`` or
``lstm``, ``--num_layers`` / ``--num_bi_layers`` as the reference's flags (:58-61; default 2 / 1), residual connections when the
model was built with them:

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


def allreduce_grads(trainer, n_tokens: int):
    """Data parallelism.  A rank's gradient is that of ITS per-token average loss over ``n_tokens`` target tokens; the
    global per-token average over all ranks' batches is sum_r(n_r * mean_r) / sum_r(n_r), so the gradients are
    weighted by the token counts: all-reduce n_r * g_r and n_r, then divide (reference loss: train_gnmt.py:332-333)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        g = trainer.grads
        cnt = torch.tensor([float(n_tokens)], dtype=torch.float32, device=g.device)
        g *= float(n_tokens)
        from .train import _grad_comm
        comm = _grad_comm(g.device)
        if comm is not None:                     # tn_allreduce_f32: RCCL behind the C-ABI
            comm.allreduce_(g)
            comm.allreduce_(cnt).wait()
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        g /= cnt


def train(data_train, data_val, data_test, model, translator, epochs: int, batch_size: int, lr: float = 1e-3,
          lr_update_factor: float = 0.5, dropout: float = 0.0, num_buckets: int = 5, test_batch_size: int = 32,
          start_epoch: int = 0, save_dir: str | None = None, seed: int = 0, log=print):
    """-> history: one dict per epoch (train loss, valid / test loss and BLEU, learning rate)."""
    enc = model.encoder
    if enc._cell_type not in ("gru", "lstm"):
        raise NotImplementedError("the training step is built for GRU / LSTM cells")
    dec = model.decoder
    if bool(getattr(enc, "_use_residual", False)) != bool(getattr(dec, "_use_residual", False)) or enc._num_layers != dec._num_layers:
        raise ValueError("encoder and decoder must agree on num_layers and use_residual (get_gnmt_encoder_decoder builds them that way, "
                         "gnmt.py:397-416)")
    params = {k: v.data for k, v in model.collect_params().items()}
    max_t = max(l[0] for l in data_train.get_data_lens())
    max_l = max(l[-1] for l in data_train.get_data_lens())
    trainer = GNMTTrainer(params, model._input_size, enc._hidden_size, model._embed_size, len(model.tgt_vocab),
                          max_batch=batch_size, max_src_len=max_t, max_tgt_len=max_l, prefix=model.prefix,
                          cell_type=enc._cell_type, num_layers=enc._num_layers, num_bi_layers=enc._num_bi_layers,
                          use_residual=bool(getattr(enc, "_use_residual", False)))
    if dropout > 0:
        trainer.set_dropout(dropout, seed)
    val_tgt = data_val.get_captions(split=True) if data_val is not None else None
    test_tgt = data_test.get_captions(split=True) if data_test is not None else None
    best_valid_bleu, history = 0.0, []
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    import torch.distributed as dist
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if ddp else (0, 1)
    for epoch_id in range(start_epoch, epochs):
        tot, nb = 0.0, 0
        # FixedBucketSampler(..., shuffle=True) of the training loader (utils/captioning.py:48-55)
        for src, tgt, svl, tvl, *_ in bucketed_batches(data_train, batch_size, num_buckets, shuffle=True, seed=seed,
                                                       epoch=epoch_id, rank=rank, world=world):
            loss = trainer.forward_backward(torch.from_numpy(src).cuda(), torch.from_numpy(svl.astype(np.int32)).cuda(),
                                            torch.from_numpy(tgt).cuda(), torch.from_numpy(tvl.astype(np.int32)).cuda())
            allreduce_grads(trainer, int((tvl.astype(np.int64) - 1).sum()))
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


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="tennis_amd train_gnmt (flags of reference train_gnmt.py:48-118)")
    p.add_argument("--model_id", default="0000")
    p.add_argument("--epochs", type=int, default=40)
    p.add_argument("--num_hidden", type=int, default=128)
    p.add_argument("--emb_size", type=int, default=100)
    p.add_argument("--dropout", type=float, default=0.2)
    p.add_argument("--num_layers", type=int, default=2)
    p.add_argument("--num_bi_layers", type=int, default=1)
    p.add_argument("--cell_type", default="gru")
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--beam_size", type=int, default=4)
    p.add_argument("--lp_alpha", type=float, default=1.0)
    p.add_argument("--lp_k", type=int, default=5)
    p.add_argument("--test_batch_size", type=int, default=32)
    p.add_argument("--num_buckets", type=int, default=5)
    p.add_argument("--tgt_max_len", type=int, default=50)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--lr_update_factor", type=float, default=0.5)
    p.add_argument("--every", type=int, default=1)
    p.add_argument("--feats_model", default=None, help="load CNN features as npy files from this model: <data_root>/features/<feats_model>/, "
                   "what `python -m tennis_amd.evaluate --save_feats --model_id <feats_model>` wrote (train_gnmt.py:116-117)")
    p.add_argument("--emb_file", default="embeddings-ex.txt", help="the word embedding file generated by train_embeddings.py, under "
                   "--data_root (train_gnmt.py:118; '' = a learned embedding of --emb_size)")
    p.add_argument("--data_root", default=None, help="the dataset directory (the reference's 'data': splits/, annotations/, features/, "
                   "the embedding file); without it the captions and features are synthetic")
    p.add_argument("--split_id", default="02")
    # the reference's remaining flags (train_gnmt.py:48-118), accepted so that its documented command lines run unchanged, e.g.
    # `python evaluate_gnmt.py --model_id 0102 --num_hidden 256 --backbone_from_id 0006 --feats_model 0006` (models/README.md:68)
    p.add_argument("--bucket_scheme", default="constant", choices=["constant"], help="bucket widths (only the reference's default is built)")
    p.add_argument("--bucket_ratio", type=float, default=0.0)
    p.add_argument("--optimizer", default="adam", choices=["adam"])
    p.add_argument("--clip", type=float, default=5.0, help="accepted; the reference defines the flag and never applies it (train_gnmt.py:305-470)")
    p.add_argument("--log_interval", type=int, default=100)
    p.add_argument("--num_gpus", type=int, default=1, help="the reference's captioner is single-GPU (train_gnmt.py:126-127); data parallelism here comes from torch.distributed")
    p.add_argument("--backbone", default="DenseNet121")
    p.add_argument("--backbone_from_id", default=None, help="frame mode only (the CNN inside the model); ignored with --feats_model, as in the reference")
    p.add_argument("--freeze_backbone", action="store_true")
    p.add_argument("--data_shape", type=int, default=512)
    p.add_argument("--feature_dim", type=int, default=1024, help="width of the pre-extracted frame features (feats_model)")
    p.add_argument("--n_points", type=int, default=64, help="synthetic source: points per split")
    p.add_argument("--root", default="models/captioning/experiments")
    return p


def load_target_embedding(flags, vocab, log=print):
    """reference train_gnmt.py:210-220: ``TokenEmbedding.from_file(data/<emb_file>)`` + ``vocab.set_embedding`` -> the table the
    target ``nn.Embedding`` is initialised with, or None (no file asked for, or a synthetic run whose directory has none)."""
    from .models.captioning.gnmt import TokenEmbedding
    if not flags.emb_file:
        return None
    path = flags.emb_file if os.path.isabs(flags.emb_file) else os.path.join(flags.data_root or "data", flags.emb_file)
    if not os.path.exists(path):
        if flags.data_root is not None:
            raise FileNotFoundError(f"--emb_file: {path} does not exist (pass --emb_file '' for a learned embedding)")
        return None                            # synthetic source: nothing on disk to read
    vocab.set_embedding(TokenEmbedding.from_file(path))
    tab = vocab.embedding.idx_to_vec
    known = int((np.abs(tab).sum(1) > 0).sum())
    log("Loaded {} x {} target embedding from {} ({} of {} vocabulary tokens found)".format(tab.shape[0], tab.shape[1], path, known, len(vocab)))
    return tab


def build(flags):
    """Datasets, model and translator as reference train_gnmt.py:120-256 assembles them (feature mode, ``--feats_model``): with
    ``--data_root`` the points, captions and per-frame ``.npy`` features come from disk (``TennisSet(captions=True, ...)``), the
    target embedding from ``--emb_file``; without it everything is synthetic."""
    from .dataset import TennisSet
    from .models.captioning.gnmt import NMTModel, get_gnmt_encoder_decoder
    from .utils.translation import BeamSearchScorer, BeamSearchTranslator
    if flags.data_root is None and flags.feats_model is not None and os.path.isdir(os.path.join("data", "splits")):
        flags.data_root = "data"           # the reference's layout relative to the working directory (dataset.py:17: root='data')
    if flags.data_root is not None and flags.feats_model is None:
        raise NotImplementedError("train_gnmt on raw frames (no --feats_model) needs the frames of every point on disk and the backbone "
                                  "inside the step; the accelerated path is the reference's feature mode (evaluate --save_feats first)")
    src = dict(root=flags.data_root, split_id=flags.split_id, feats_model=flags.feats_model) if flags.data_root is not None else dict(root=None)
    syn = {} if flags.data_root is not None else dict(feature_dim=flags.feature_dim)
    data_train = TennisSet(captions=True, split="train", every=flags.every, max_cap_len=flags.tgt_max_len,
                           **src, **syn, **({} if flags.data_root is not None else dict(n_points=flags.n_points)))
    data_val = TennisSet(captions=True, split="val", every=flags.every, vocab=data_train.vocab, inference=True,
                         **src, **syn, **({} if flags.data_root is not None else dict(n_points=max(4, flags.n_points // 4))))
    data_test = TennisSet(captions=True, split="test", every=flags.every, vocab=data_train.vocab, inference=True,
                          **src, **syn, **({} if flags.data_root is not None else dict(n_points=max(4, flags.n_points // 4))))
    feature_dim = data_train[0][0].shape[1] if len(data_train) else flags.feature_dim      # the width evaluate --save_feats wrote
    tgt_embed = load_target_embedding(flags, data_train.vocab)
    enc, dec = get_gnmt_encoder_decoder(cell_type=flags.cell_type, hidden_size=flags.num_hidden, dropout=flags.dropout,
                                        num_layers=flags.num_layers, num_bi_layers=flags.num_bi_layers)
    model = NMTModel(src_vocab=None, tgt_vocab=data_train.vocab, encoder=enc, decoder=dec, embed_size=flags.emb_size,
                     prefix="gnmt_", input_size=feature_dim, tgt_embed=tgt_embed)                  # train_gnmt.py:228-229
    model.initialize()
    translator = BeamSearchTranslator(model=model, beam_size=flags.beam_size,
                                      scorer=BeamSearchScorer(alpha=flags.lp_alpha, K=flags.lp_k),
                                      max_length=flags.tgt_max_len + 100)                     # train_gnmt.py:250-252
    return data_train, data_val, data_test, model, translator


def main(argv=None):
    flags = build_parser().parse_args(argv)
    data_train, data_val, data_test, model, translator = build(flags)
    save_dir = os.path.join(flags.root, flags.model_id)
    os.makedirs(save_dir, exist_ok=True)
    from .captions import write_sentences
    write_sentences(data_val.get_captions(split=True), os.path.join(save_dir, "val_gt.txt"))         # train_gnmt.py:205-208
    write_sentences(data_test.get_captions(split=True), os.path.join(save_dir, "test_gt.txt"))
    # resume (train_gnmt.py:232-244): the newest NNNN.params of the model id, valid_best.params aside
    start_epoch = 0
    files = sorted((f for f in os.listdir(save_dir) if f.endswith(".params") and f != "valid_best.params"), reverse=True)
    if files:
        start_epoch = int(files[0].split(".")[0]) + 1
        model.load_parameters(os.path.join(save_dir, files[0]))
        print("Loaded model params: {}".format(os.path.join(save_dir, files[0])))
    hist = train(data_train, data_val, data_test, model, translator, flags.epochs, flags.batch_size, lr=flags.lr,
                 lr_update_factor=flags.lr_update_factor, dropout=flags.dropout, num_buckets=flags.num_buckets,
                 test_batch_size=flags.test_batch_size, start_epoch=start_epoch, save_dir=save_dir)
    if not hist:
        print("[Finished] nothing to do: {} epochs are on disk".format(start_epoch))
        return 0
    print("[Finished] best valid bleu={:.2f}".format(100 * max(h.get("valid_bleu", 0.0) for h in hist)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
