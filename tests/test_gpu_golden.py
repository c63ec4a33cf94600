"""-m gpu: the HIP path against the COMMITTED fixtures tests/golden/oracle_*.npz (SURVEY §8c list; produced by the CPU
oracle on seeded inputs, tests/golden/make_oracle_fixtures.py) - the same bars as the live-oracle parity tests:
features / logits 1e-3, recurrent layers and the decoder step 1e-4, caption token ids exact."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mk():
    spec = importlib.util.spec_from_file_location("make_oracle_fixtures", os.path.join(GOLD, "make_oracle_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)      # imports oracle/ for its input generators only (tests may)
    return m


def test_densenet_features_and_logits(mk, report):
    from tennis_amd.engine import Dense, DenseNet121Features
    d = np.load(os.path.join(GOLD, "oracle_densenet121_224_b2.npz"))
    from tennis_amd import weights as W
    p, _ = mk.densenet_inputs()
    enc = DenseNet121Features(p, 224, max_batch=2)
    feats = enc(torch.from_numpy(W.synthetic_frames_u8(2, 224, 1234)).cuda())      # the decoded frames the fixture's normalised input was made from
    logits = Dense(p["framemodel0_dense0_weight"], p["framemodel0_dense0_bias"])(feats).cpu().numpy()
    ef, el = float(np.abs(feats.cpu().numpy() - d["feats"]).max()), float(np.abs(logits - d["logits"]).max())
    report["golden_densenet_feat_maxabs_err"], report["golden_densenet_logit_maxabs_err"] = ef, el
    assert ef < 1e-3 and el < 1e-3, (ef, el)
    for tap in ("pool0", "stage1", "trans1", "stage2", "trans2", "stage3", "trans3", "stage4"):
        got = enc.read_tap(tap, 2)
        assert abs(float(got.mean()) - float(d[f"tap_{tap}_mean"])) < 2e-3 * max(1.0, float(d[f"tap_{tap}_absmax"])), tap


@pytest.mark.parametrize("mode", ["gru", "lstm"])
def test_birnn(mk, mode):
    from tennis_amd.engine import BiRNN
    g = np.load(os.path.join(GOLD, f"oracle_bi{mode}_b2_t8_f1024.npz"))
    p, x, vl = mk.rnn_inputs(mode)
    net = BiRNN(mode, 1024, 128, p, "rnn_", True, max_rows=16)
    seq, hl, _ = net(torch.from_numpy(x).cuda(), None, True)
    assert np.abs(seq.cpu().numpy() - g["seq"]).max() < 1e-4
    assert np.abs(hl[0].cpu().numpy() - g["h_fwd"]).max() < 1e-4 and np.abs(hl[1].cpu().numpy() - g["h_bwd"]).max() < 1e-4
    seq, hl, _ = net(torch.from_numpy(x).cuda(), torch.from_numpy(vl).cuda(), True)
    assert np.abs(seq.cpu().numpy() - g["seq_ragged"]).max() < 1e-4 and np.abs(hl[1].cpu().numpy() - g["h_bwd_ragged"]).max() < 1e-4


def test_gnmt_step_and_beam_trace(mk):
    from tennis_amd.engine import GNMTCaptioner
    c = mk.GN
    p, src, vl = mk.gnmt_inputs()
    cap = GNMTCaptioner(p, c["F"], c["H"], c["E"], c["V"], beam=c["beam"], max_length=c["max_length"], max_batch=c["B"],
                        max_src_len=c["T"])
    mem = cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(vl).cuda()).cpu().numpy()
    g = np.load(os.path.join(GOLD, "oracle_gnmt_step.npz"))
    assert np.abs(mem - g["mem"]).max() < 1e-4
    # one teacher-forced step from the encoder state = the fixture's decoder step (log-softmax of its logits)
    tgt = torch.from_numpy(g["tokens"].astype(np.int32).reshape(-1, 1)).cuda()
    logits = cap.decode_seq(tgt).cpu().numpy()[:, 0]
    m = logits.max(axis=1, keepdims=True)
    logp = logits - m - np.log(np.exp(logits - m).sum(axis=1, keepdims=True))
    assert np.abs(logp - g["logp"]).max() < 1e-4
    t = np.load(os.path.join(GOLD, "oracle_beam_trace.npz"))
    cap.encode(torch.from_numpy(src).cuda(), torch.from_numpy(vl).cuda())
    s, sc, vlen = cap.beam_search(2, 3, 1.0, 5.0)
    assert np.array_equal(s.cpu().numpy(), t["samples"]) and np.array_equal(vlen.cpu().numpy(), t["valid_length"])
    assert np.abs(sc.cpu().numpy() - t["scores"]).max() < 1e-4
