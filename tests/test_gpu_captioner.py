"""-m gpu: GNMT encoder + beam search through the C ABI vs the CPU oracle: caption token ids
must be equal (BASELINE.json north_star), scores within 1e-4."""
import numpy as np
import pytest
import torch

from oracle import gnmt_np as gn

pytestmark = pytest.mark.gpu


def _case(seed, B, T, F, H, E, V, beam, max_length, eos_bias=0.0, proj_scale=1.0, cell="gru", nl=2, nbi=1, res=False):
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    p = W.make_gnmt_weights(seed, cell, F, H, E, V, num_layers=nl, num_bi_layers=nbi)
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * proj_scale).astype(np.float32)   # peakier word distribution
    p["gnmt_tgt_proj_bias"][3] += eos_bias          # steer how early <eos> wins
    rng = np.random.default_rng(seed)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    vl = rng.integers(max(1, T // 3), T + 1, B).astype(np.int32)
    vl[0] = T
    cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=max_length, max_batch=B, max_src_len=T, cell_type=cell,
                        num_layers=nl, num_bi_layers=nbi, use_residual=res)
    mem = cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(vl).cuda()).cpu().numpy()
    samples, scores, vlen = cap.beam_search(2, 3, 1.0, 5.0)
    rmem, rstates = gn.encoder(src, vl, p, cell, H, num_layers=nl, num_bi_layers=nbi, use_residual=res)
    dec = gn.Decoder(p, H, num_layers=nl, cell=cell, use_residual=res)
    rs, rsc, rvl = gn.beam_search(dec, rmem, rstates, vl, 2, 3, beam, 1.0, 5, max_length)
    return (mem, samples.cpu().numpy(), scores.cpu().numpy(), vlen.cpu().numpy()), (rmem, rs, rsc, rvl)


@pytest.mark.parametrize("cfg", [
    dict(seed=1, B=3, T=11, F=24, H=16, E=12, V=30, beam=4, max_length=12),                    # runs to max_length
    dict(seed=2, B=5, T=23, F=64, H=32, E=20, V=40, beam=4, max_length=40, eos_bias=1.2),      # every beam finishes early
    dict(seed=2, B=5, T=23, F=64, H=32, E=20, V=40, beam=4, max_length=40, proj_scale=40.0),   # finished + unfinished beams mixed
    dict(seed=3, B=4, T=60, F=1024, H=128, E=100, V=254, beam=5, max_length=30, proj_scale=30.0),  # config C5 shape
    dict(seed=4, B=3, T=13, F=32, H=16, E=12, V=30, beam=4, max_length=14, cell="lstm"),                    # LSTM cells
    dict(seed=5, B=5, T=23, F=64, H=32, E=20, V=40, beam=4, max_length=40, proj_scale=40.0, cell="lstm"),
    # the other instantiations of the step kernels: beam rows padded to 4 / 8 / 16, source longer than one pass of
    # the attention kernel (T > 256), vocabulary wider than one pass of the projection (V > 256), hidden % 64 != 0
    dict(seed=6, B=3, T=17, F=32, H=16, E=12, V=30, beam=1, max_length=16, proj_scale=30.0),
    dict(seed=7, B=3, T=17, F=32, H=24, E=10, V=30, beam=3, max_length=16, proj_scale=30.0),
    dict(seed=8, B=2, T=300, F=48, H=32, E=16, V=300, beam=7, max_length=20, proj_scale=40.0),
    dict(seed=9, B=2, T=40, F=48, H=32, E=16, V=61, beam=10, max_length=20, proj_scale=40.0, cell="lstm"),
    # --num_layers / --num_bi_layers other than the flag defaults (train_gnmt.py:72-75), use_residual (gnmt.py:155-157,394-395)
    dict(seed=10, B=3, T=19, F=40, H=32, E=16, V=40, beam=4, max_length=24, proj_scale=30.0, nl=3, nbi=1),
    dict(seed=11, B=3, T=19, F=40, H=32, E=16, V=40, beam=5, max_length=24, proj_scale=30.0, nl=4, nbi=2, cell="lstm"),
    dict(seed=12, B=4, T=15, F=40, H=32, E=16, V=40, beam=4, max_length=24, proj_scale=30.0, nl=4, nbi=1, res=True),
    dict(seed=13, B=3, T=15, F=40, H=32, E=16, V=40, beam=4, max_length=24, proj_scale=30.0, nl=2, nbi=1, res=True, cell="lstm"),
    dict(seed=14, B=3, T=15, F=40, H=32, E=16, V=40, beam=3, max_length=20, proj_scale=30.0, nl=3, nbi=0, res=True),
    # shapes whose 16-way partial sums do not fit the step kernels' LDS (the split is lowered): wide beam x full vocabulary x H = 256,
    # a source of 2500 steps
    dict(seed=15, B=2, T=30, F=64, H=256, E=100, V=254, beam=10, max_length=12, proj_scale=30.0),
    dict(seed=16, B=1, T=2500, F=16, H=32, E=16, V=40, beam=3, max_length=8, proj_scale=30.0),
    # 12 clips: eight of them through the XCD-aware workgroup -> row mapping of the attention kernel, four behind it
    dict(seed=17, B=12, T=21, F=32, H=32, E=16, V=40, beam=5, max_length=14, proj_scale=30.0),
])
def test_beam_search_matches_oracle(cfg, report):
    (mem, s, sc, vl), (rmem, rs, rsc, rvl) = _case(**cfg)
    assert np.abs(mem - rmem).max() < 1e-4
    report[f"gnmt_{cfg.get('cell', 'gru')}_seed{cfg['seed']}_nl{cfg.get('nl', 2)}_ps{cfg.get('proj_scale', 1.0)}_score_maxabs_err"] = float(np.abs(sc - rsc).max())
    assert s.shape == rs.shape, (s.shape, rs.shape)
    assert np.array_equal(vl, rvl)
    assert np.array_equal(s, rs)                       # caption token ids equal
    assert np.abs(sc - rsc).max() < 1e-4
    assert np.all(np.diff(sc, axis=1) <= 1e-6)         # scores[i, :] descending (translation.py:69-70)
    assert np.all(s[:, :, 0] == 2)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_translator_surface(cell):
    """NMTModel + BeamSearchTranslator + ids -> tokens as in reference train_gnmt.py:287-300."""
    from tennis_amd.models.captioning.gnmt import NMTModel, Vocab, get_gnmt_encoder_decoder
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    words = "the player serves near far left right a forehand backhand return in out".split()
    vocab = Vocab({w: i + 1 for i, w in enumerate(words)})
    enc, dec = get_gnmt_encoder_decoder(cell_type=cell, hidden_size=32, dropout=0.2, num_layers=2, num_bi_layers=1)
    model = NMTModel(src_vocab=None, tgt_vocab=vocab, encoder=enc, decoder=dec, embed_size=12, prefix="gnmt_",
                     input_size=48)
    model.initialize()
    tr = BeamSearchTranslator(model=model, beam_size=4, scorer=BeamSearchScorer(alpha=1.0, K=5), max_length=20)
    src = np.abs(np.random.default_rng(0).normal(0, 1, (3, 9, 48))).astype(np.float32)
    samples, scores, vlen = tr.translate(src, np.array([9, 5, 7], np.float32))
    best = samples[:, 0, :].cpu().numpy()
    vl0 = vlen[:, 0].cpu().numpy()
    sents = [[vocab.idx_to_token[e] for e in best[i][1:(vl0[i] - 1)]] for i in range(3)]
    assert len(sents) == 3 and all(isinstance(w, str) for s in sents for w in s)
    p = {k: v.data for k, v in model.collect_params().items()}
    rmem, rstates = gn.encoder(src, np.array([9, 5, 7]), p, cell, 32)
    rs, _, rvl = gn.beam_search(gn.Decoder(p, 32, cell=cell), rmem, rstates, np.array([9, 5, 7]), 2, 3, 4, 1.0, 5, 20)
    assert sents == gn.ids_to_sentences(rs, rvl, vocab.idx_to_token)


def test_teacher_forcing_and_loss(report):
    """model(src, tgt[:, :-1]) logits and MaskedSoftmaxCELoss (train_gnmt.py:280-281) vs the oracle."""
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner, masked_softmax_ce
    B, T, F, H, E, V, L = 4, 17, 48, 32, 16, 50, 9
    p = W.make_gnmt_weights(11, "gru", F, H, E, V)
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * 20).astype(np.float32)
    rng = np.random.default_rng(11)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    svl = np.array([17, 9, 12, 5], np.int32)
    tgt = rng.integers(4, V, (B, L)).astype(np.int32)
    tvl = np.array([9, 4, 7, 9], np.int32)
    cap = GNMTCaptioner(p, F, H, E, V, beam=4, max_length=20, max_batch=B, max_src_len=T)
    cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(svl).cuda())
    logits = cap.decode_seq(torch.from_numpy(tgt[:, :-1]).cuda())
    loss = masked_softmax_ce(logits, torch.from_numpy(tgt[:, 1:]).cuda(), torch.from_numpy(tvl - 1).cuda()).cpu().numpy()
    rmem, rstates = gn.encoder(src, svl, p, "gru", H)
    rl = gn.decode_seq(gn.Decoder(p, H), rmem, rstates, svl, tgt[:, :-1])
    rloss = gn.masked_softmax_ce(rl, tgt[:, 1:], tvl - 1)
    report["gnmt_teacher_forced_logits_maxabs_err"] = float(np.abs(logits.cpu().numpy() - rl).max())
    assert np.abs(logits.cpu().numpy() - rl).max() < 1e-4
    assert np.abs(loss - rloss).max() < 1e-5


@pytest.mark.parametrize("nl,nbi,res,cell", [(3, 1, False, "gru"), (4, 2, True, "lstm"), (2, 1, True, "gru")])
def test_teacher_forcing_layer_counts_and_residual(report, nl, nbi, res, cell):
    """decode_seq logits with num_layers / num_bi_layers / use_residual other than the reference's defaults vs the oracle."""
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    B, T, F, H, E, V, L = 3, 13, 40, 32, 16, 50, 8
    p = W.make_gnmt_weights(20 + nl, cell, F, H, E, V, num_layers=nl, num_bi_layers=nbi)
    rng = np.random.default_rng(nl)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    svl = np.array([13, 6, 9], np.int32)
    tgt = rng.integers(4, V, (B, L)).astype(np.int32)
    cap = GNMTCaptioner(p, F, H, E, V, beam=2, max_length=12, max_batch=B, max_src_len=T, cell_type=cell, num_layers=nl,
                        num_bi_layers=nbi, use_residual=res)
    mem = cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(svl).cuda()).cpu().numpy()
    logits = cap.decode_seq(torch.from_numpy(tgt).cuda()).cpu().numpy()
    rmem, rstates = gn.encoder(src, svl, p, cell, H, num_layers=nl, num_bi_layers=nbi, use_residual=res)
    rl = gn.decode_seq(gn.Decoder(p, H, num_layers=nl, cell=cell, use_residual=res), rmem, rstates, svl, tgt)
    assert np.abs(mem - rmem).max() < 1e-4
    err = float(np.abs(logits - rl).max())
    report[f"gnmt_tf_nl{nl}_nbi{nbi}_res{int(res)}_{cell}_logits_maxabs_err"] = err
    assert err < 1e-4, err
    with pytest.raises(RuntimeError):       # every layer bidirectional: the memory would be 2H wide (tn_gnmt_create_ex)
        GNMTCaptioner(W.make_gnmt_weights(1, cell, F, H, E, V, num_layers=2, num_bi_layers=2), F, H, E, V, num_layers=2, num_bi_layers=2)


def test_captioning_evaluate_driver():
    """CaptionSet + bucketed Pad batches + evaluate() (train_gnmt.py:264-302): sentences come back in
    dataset order and equal the oracle's."""
    from tennis_amd.captions import CaptionSet, bucketed_batches, evaluate
    from tennis_amd.models.captioning.gnmt import NMTModel, get_gnmt_encoder_decoder
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    train = CaptionSet(split="train", n_points=12, feature_dim=64, mean_frames=12, max_cap_len=50)
    test = CaptionSet(split="test", n_points=7, feature_dim=64, mean_frames=12, vocab=train.vocab, inference=True)
    assert test.vocab is train.vocab and test[0][1][0] == 2 and test[0][1][-1] == 3
    enc, dec = get_gnmt_encoder_decoder(cell_type="gru", hidden_size=32, num_layers=2, num_bi_layers=1)
    model = NMTModel(src_vocab=None, tgt_vocab=train.vocab, encoder=enc, decoder=dec, embed_size=16, prefix="gnmt_",
                     input_size=64)
    model.initialize()
    tr = BeamSearchTranslator(model=model, beam_size=4, scorer=BeamSearchScorer(alpha=1.0, K=5), max_length=12)
    loss, sents = evaluate(bucketed_batches(test, batch_size=3), model, tr, train)
    assert len(sents) == 7 and all(s is not None for s in sents) and np.isfinite(loss)
    p = {k: v.data for k, v in model.collect_params().items()}
    for i in range(7):                                   # per-clip oracle, no padding involved
        x, cap, tl, cl, idx = test[i]
        rmem, rst = gn.encoder(x[None], np.array([tl]), p, "gru", 32)
        rs, _, rvl = gn.beam_search(gn.Decoder(p, 32), rmem, rst, np.array([tl]), 2, 3, 4, 1.0, 5, 12)
        assert sents[i] == gn.ids_to_sentences(rs, rvl, train.vocab.idx_to_token)[0]


def test_captioning_pipeline_from_disk_through_the_drivers(tmp_path, report):
    """the pipeline the reference runs (train_gnmt.py:116-118,196-218; evaluate.py --save_feats) started from
    the drivers' command lines on a tiny copy of the reference's data/ directory:
        python -m tennis_amd.evaluate --save_feats --model_id 0042 --split {train,val,test}     frames (JPEG) -> features (.npy)
        python -m tennis_amd.train_gnmt --data_root data --feats_model 0042                       features + captions + embeddings -> captioner
        python -m tennis_amd.evaluate_gnmt ...                                                    best parameters -> loss / BLEU / sentences"""
    import os
    from tennis_amd import evaluate as ev, evaluate_gnmt as eg, train_gnmt as tg
    from tennis_amd.dataset import TennisSet
    from tools import tiny_dataset as td
    root, exp = str(tmp_path / "data"), str(tmp_path / "exp")
    info = td.write(root, np.random.default_rng(8))
    for split in ("train", "val", "test"):
        assert ev.main(["--root", root, "--model_id", "0042", "--save_feats", "--split", split, "--batch_size", "16",
                        "--exp_root", str(tmp_path / "vexp"), "--num_workers", "0"]) == 0
    pid, v, a, b, cap = info["points"]["test"][0]
    f = np.load(TennisSet.get_feature_path(os.path.join(root, "features", "0042"), v, a))
    assert f.shape == (1024,) and f.dtype == np.float32 and np.isfinite(f).all() and f.std() > 0
    common = ["--data_root", root, "--feats_model", "0042", "--model_id", "cap1", "--root", exp, "--num_hidden", "16",
              "--tgt_max_len", "12", "--beam_size", "2", "--test_batch_size", "2", "--num_buckets", "2"]
    assert tg.main(common + ["--epochs", "3", "--batch_size", "2", "--dropout", "0.0", "--lr", "0.01"]) == 0
    files = sorted(os.listdir(os.path.join(exp, "cap1")))
    assert {"0000.params", "0002.params", "val_gt.txt", "test_gt.txt", "epoch0_valid_out.txt", "epoch2_test_out.txt"} <= set(files), files
    assert open(os.path.join(exp, "cap1", "test_gt.txt")).read().split("\n")[:2] == td.CAPTIONS["test"]
    # a second call resumes behind the newest epoch file (train_gnmt.py:232-244): one more epoch, not three
    assert tg.main(common + ["--epochs", "4", "--batch_size", "2", "--dropout", "0.0", "--lr", "0.01"]) == 0
    assert "0003.params" in os.listdir(os.path.join(exp, "cap1")) and "epoch3_test_out.txt" in os.listdir(os.path.join(exp, "cap1"))
    out = eg.main(common)
    assert set(out) == {"valid", "test"} and all(np.isfinite(l) and 0.0 <= bl <= 1.0 for l, bl in out.values())
    assert open(os.path.join(exp, "cap1", "best_test_out.txt")).read().count("\n") == 2       # one line per test point (a barely trained model may emit <eos> at once)
    # the model the drivers built: 1024-d frame features in, the embedding file's width as embed size, its rows as the table's start
    d_tr, _, _, model, _ = tg.build(tg.build_parser().parse_args(common))
    assert model._input_size == 1024 and model._embed_size == 12 and len(d_tr.vocab) == 4 + len({w for c in td.CAPTIONS["train"] for w in c.split()})
    report["captioning_pipeline_test_loss"] = float(out["test"][0])


def test_frame_mode_source_embedding(report):
    """Frame-mode captioner (reference train_gnmt.py:148-170): ``src_embed = TimeDistributed(FrameModel(...).backbone)`` -
    the clip's FRAMES go into the model; translations and teacher-forced logits equal those of the feature-mode model
    fed with the backbone's features of the same frames."""
    from tennis_amd import weights as W
    from tennis_amd.model_zoo import get_model
    from tennis_amd.models.captioning.gnmt import NMTModel, Vocab, get_gnmt_encoder_decoder
    from tennis_amd.models.vision.definitions import FrameModel
    from tennis_amd.utils.layers import TimeDistributed
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    vocab = Vocab({f"w{i}": 30 - i for i in range(26)})
    cnn_model = FrameModel(get_model("DenseNet121", pretrained=True, seed=0).features, 11)      # train_gnmt.py:150-151
    src_embed = TimeDistributed(cnn_model.backbone)                                             # :168-170
    mk = lambda se: NMTModel(src_vocab=None, tgt_vocab=vocab, encoder=get_gnmt_encoder_decoder(cell_type="gru", hidden_size=32)[0],
                             decoder=get_gnmt_encoder_decoder(cell_type="gru", hidden_size=32)[1], embed_size=16, prefix="gnmt_",
                             src_embed=se, input_size=1024, seed=5)
    frame_model, feat_model = mk(src_embed), mk(None)
    frame_model.initialize(); feat_model.initialize()
    B, T = 2, 5
    frames = torch.from_numpy(W.synthetic_frames_u8(B * T, 224).reshape(B, T, 224, 224, 3)).cuda()    # decoded uint8 clips
    vl = np.array([5, 3], np.float32)
    feats = cnn_model.backbone(frames.reshape(B * T, 224, 224, 3)).reshape(B, T, 1024)
    assert torch.equal(frame_model.embed_source(frames), feats)
    out = []
    for m, src in ((frame_model, frames), (feat_model, feats)):
        tr = BeamSearchTranslator(model=m, beam_size=3, scorer=BeamSearchScorer(alpha=1.0, K=5), max_length=10)
        out.append([t.cpu().numpy() for t in tr.translate(src, vl)])
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    report["frame_mode_captioner_equals_feature_mode"] = True


def test_full_size_c5_properties(report):
    """BASELINE.json config C5 size (32 clips, T=214, F=1024, H=256, E=100, V=254, beam 5, max_len 150) — too long
    for the numpy oracle, so size-independent properties: (1) batch invariance: clip i decoded in the batch of 32
    gives the same token ids / lengths as decoded alone; (2) BOS first, EOS at valid_length-1, -1 padding behind;
    (3) scores descending over beams; (4) beam 1 == greedy argmax of the teacher-forced logits."""
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
    p = W.make_gnmt_weights(9, "gru", F, H, E, V)
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * 30.0).astype(np.float32)
    p["gnmt_tgt_proj_bias"][3] += 0.5
    rng = np.random.default_rng(9)
    src = torch.from_numpy((np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)).cuda()
    vl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).cuda()
    cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T)
    cap.encode(src, vl)
    s, sc, svl = [x.cpu().numpy() for x in cap.beam_search(2, 3, 1.0, 5.0)]
    assert (s[:, :, 0] == 2).all() and np.all(np.diff(sc, axis=1) <= 1e-6)
    L = s.shape[2]
    for b in range(B):
        for k in range(beam):
            n = int(svl[b, k])
            assert s[b, k, n - 1] == 3 and (s[b, k, n:] == -1).all() and (s[b, k, 1:n - 1] >= 0).all()
    one = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=1, max_src_len=T)
    for b in (0, 7, 31):
        one.encode(src[b:b + 1].contiguous(), vl[b:b + 1].contiguous())
        s1, sc1, v1 = [x.cpu().numpy() for x in one.beam_search(2, 3, 1.0, 5.0)]
        n = int(v1[0, 0])
        assert np.array_equal(v1[0], svl[b]) and np.array_equal(s1[0, 0, :n], s[b, 0, :n])
    g = GNMTCaptioner(p, F, H, E, V, beam=1, max_length=ml, max_batch=B, max_src_len=T)
    g.encode(src, vl)
    s1, _, v1 = [x.cpu().numpy() for x in g.beam_search(2, 3, 1.0, 5.0)]
    n = int(v1.max())
    tgt = np.where(s1[:, 0, :n - 1] < 0, 3, s1[:, 0, :n - 1]).astype(np.int32)      # feed the greedy path back in
    logits = g.decode_seq(torch.from_numpy(tgt).cuda()).cpu().numpy()
    for b in range(B):
        m = min(int(v1[b, 0]) - 1, ml)       # predictions for positions 1..m (an <eos> forced at max_length is no argmax)
        assert np.array_equal(logits[b, :m].argmax(-1), s1[b, 0, 1:m + 1])
    report["gnmt_c5_full_size_properties"] = True


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_full_size_c5_against_the_oracle(cell, report):
    """BASELINE.json config C5 at FULL size (32 clips, T = 214, F = 1024, H = 256, E = 100, V = 254, beam 5, 150 steps) against the numpy
    oracle (4 - 8 s of CPU: the oracle was thought too slow for this until round 4 timed it).  GRU (the reference's flag default):
    every token id of every beam of every clip, every length, scores to 1e-4.  LSTM: the same for the best beam of every clip and
    for every other beam unless the oracle itself is undecided there at float32 precision (on one clip the fifth beam is a near
    tie at an intermediate step: the oracle run on float64 copies of its inputs picks the device's hypothesis, the float32 run the
    other one) - every beam of the device equals one of the oracle's two runs."""
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
    p = W.make_gnmt_weights(9, cell, F, H, E, V)
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * 30.0).astype(np.float32)
    p["gnmt_tgt_proj_bias"][3] += 0.5
    rng = np.random.default_rng(9)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    vl = np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)
    cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T, cell_type=cell)
    mem_d = cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(vl).cuda()).cpu().numpy()
    s, sc, svl = [x.cpu().numpy() for x in cap.beam_search(2, 3, 1.0, 5.0)]
    mem, states = gn.encoder(src, vl, p, cell, H)
    rs, rsc, rv = gn.beam_search(gn.Decoder(p, H, cell=cell), mem, states, vl, 2, 3, beam=beam, max_length=ml)
    assert np.abs(mem_d - mem).max() < 1e-4
    assert s.shape == rs.shape and np.array_equal(svl, rv) and np.abs(sc - rsc).max() < 1e-4
    eq = s == rs
    report[f"gnmt_c5_full_size_{cell}_ids_equal"] = float(eq.mean())
    assert eq[:, 0].all()                                    # the caption that is written out: the best beam
    if cell == "gru":
        assert eq.all()
    else:
        # No allowance by count.  A beam may differ from the oracle only where the ORACLE ITSELF is undecided at
        # float32 precision: the same oracle fed float64 copies of the parameters and features (its matrix products then
        # accumulate in double before the cast) ranks one hypothesis of one clip differently from its float32 run - a near tie at
        # an intermediate step that the projection's accumulation order decides; gluonnlp's topk has no tie rule to mirror beyond
        # "larger score first", which both sides implement.  Every beam of the device must equal one of the two runs, ids and all.
        p64 = {k: v.astype(np.float64) for k, v in p.items()}
        mem64, states64 = gn.encoder(src.astype(np.float64), vl, p64, cell, H)
        rs64, rsc64, rv64 = gn.beam_search(gn.Decoder(p64, H, cell=cell), mem64, states64, vl, 2, 3, beam=beam, max_length=ml)
        either = eq.all(axis=2) | (s == rs64).all(axis=2)
        report[f"gnmt_c5_full_size_{cell}_near_tie_beams"] = int((~eq.all(axis=2)).sum())
        assert either.all(), list(zip(*np.nonzero(~either)))
        assert eq.mean() > 0.999


def test_teacher_forcing_lstm_cells():
    """decode_seq with cell_type='lstm' (h and c carried from the encoder, gnmt.py:224-252) vs the oracle."""
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    B, T, F, H, E, V, L = 3, 11, 24, 16, 8, 30, 7
    p = W.make_gnmt_weights(13, "lstm", F, H, E, V)
    rng = np.random.default_rng(13)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    svl = np.array([11, 6, 9], np.int32)
    tgt = rng.integers(4, V, (B, L)).astype(np.int32)
    cap = GNMTCaptioner(p, F, H, E, V, beam=2, max_length=10, max_batch=B, max_src_len=T, cell_type="lstm")
    cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(svl).cuda())
    logits = cap.decode_seq(torch.from_numpy(tgt).cuda()).cpu().numpy()
    rmem, rstates = gn.encoder(src, svl, p, "lstm", H)
    rl = gn.decode_seq(gn.Decoder(p, H, cell="lstm"), rmem, rstates, svl, tgt)
    assert np.abs(logits - rl).max() < 1e-4
