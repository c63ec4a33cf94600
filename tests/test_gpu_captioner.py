"""-m gpu: GNMT encoder + beam search through the C ABI vs the CPU oracle: caption token ids
must be equal (BASELINE.json north_star), scores within 1e-4."""
import numpy as np
import pytest
import torch

from oracle import gnmt_np as gn

pytestmark = pytest.mark.gpu


def _case(seed, B, T, F, H, E, V, beam, max_length, eos_bias=0.0, proj_scale=1.0):
    from tennis_amd import weights as W
    from tennis_amd.engine import GNMTCaptioner
    p = W.make_gnmt_weights(seed, "gru", F, H, E, V)
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * proj_scale).astype(np.float32)   # peakier word distribution
    p["gnmt_tgt_proj_bias"][3] += eos_bias          # steer how early <eos> wins
    rng = np.random.default_rng(seed)
    src = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    vl = rng.integers(max(1, T // 3), T + 1, B).astype(np.int32)
    vl[0] = T
    cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=max_length, max_batch=B, max_src_len=T)
    mem = cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(vl).cuda()).cpu().numpy()
    samples, scores, vlen = cap.beam_search(2, 3, 1.0, 5.0)
    rmem, rstates = gn.encoder(src, vl, p, "gru", H)
    dec = gn.Decoder(p, H)
    rs, rsc, rvl = gn.beam_search(dec, rmem, rstates, vl, 2, 3, beam, 1.0, 5, max_length)
    return (mem, samples.cpu().numpy(), scores.cpu().numpy(), vlen.cpu().numpy()), (rmem, rs, rsc, rvl)


@pytest.mark.parametrize("cfg", [
    dict(seed=1, B=3, T=11, F=24, H=16, E=12, V=30, beam=4, max_length=12),                    # runs to max_length
    dict(seed=2, B=5, T=23, F=64, H=32, E=20, V=40, beam=4, max_length=40, eos_bias=1.2),      # every beam finishes early
    dict(seed=2, B=5, T=23, F=64, H=32, E=20, V=40, beam=4, max_length=40, proj_scale=40.0),   # finished + unfinished beams mixed
    dict(seed=3, B=4, T=60, F=1024, H=128, E=100, V=254, beam=5, max_length=30, proj_scale=30.0),  # config C5 shape
])
def test_beam_search_matches_oracle(cfg, report):
    (mem, s, sc, vl), (rmem, rs, rsc, rvl) = _case(**cfg)
    assert np.abs(mem - rmem).max() < 1e-4
    report[f"gnmt_seed{cfg['seed']}_ps{cfg.get('proj_scale', 1.0)}_score_maxabs_err"] = float(np.abs(sc - rsc).max())
    assert s.shape == rs.shape, (s.shape, rs.shape)
    assert np.array_equal(vl, rvl)
    assert np.array_equal(s, rs)                       # caption token ids equal
    assert np.abs(sc - rsc).max() < 1e-4
    assert np.all(np.diff(sc, axis=1) <= 1e-6)         # scores[i, :] descending (translation.py:69-70)
    assert np.all(s[:, :, 0] == 2)


def test_translator_surface():
    """NMTModel + BeamSearchTranslator + ids -> tokens as in reference train_gnmt.py:287-300."""
    from tennis_amd.models.captioning.gnmt import NMTModel, Vocab, get_gnmt_encoder_decoder
    from tennis_amd.utils.translation import BeamSearchScorer, BeamSearchTranslator
    words = "the player serves near far left right a forehand backhand return in out".split()
    vocab = Vocab({w: i + 1 for i, w in enumerate(words)})
    enc, dec = get_gnmt_encoder_decoder(cell_type="gru", hidden_size=32, dropout=0.2, num_layers=2, num_bi_layers=1)
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
    rmem, rstates = gn.encoder(src, np.array([9, 5, 7]), p, "gru", 32)
    rs, _, rvl = gn.beam_search(gn.Decoder(p, 32), rmem, rstates, np.array([9, 5, 7]), 2, 3, 4, 1.0, 5, 20)
    assert sents == gn.ids_to_sentences(rs, rvl, vocab.idx_to_token)
