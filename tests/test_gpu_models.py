"""-m gpu: the reference's model surface (FrameModel / TemporalPooling / CNNRNN /
save_features / evaluate_model) running on the HIP library vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import densenet_np as dn
from oracle import vision_np as vn

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def zoo():
    from tennis_amd import weights as W
    from tennis_amd.model_zoo import get_model
    from tennis_amd.models.vision.definitions import FrameModel
    backbone = get_model("DenseNet121", pretrained=True, seed=0).features
    fm = FrameModel(backbone, 11, prefix="framemodel0_")
    fm.initialize(); fm.hybridize()
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    fm.set_params(p)
    return fm, p


def _frames(n, seed=1234):
    from tennis_amd import weights as W
    x = W.normalize_to_nchw_f32(W.synthetic_frames_u8(n, 224, seed))
    return x.astype(np.float16).astype(np.float32)   # fp16-representable NCHW fp32 (reference layout)


def test_frame_model_logits(zoo, report):
    fm, p = zoo
    x = _frames(2)
    got = fm(x).cpu().numpy()
    ref = vn.frame_model(x, p)
    e = float(np.abs(got - ref).max())
    report["FrameModel_logits_maxabs_err"] = e
    assert got.shape == (2, 11) and e < TOL
    feats = fm.backbone(x).cpu().numpy()
    assert feats.shape == (2, 1024)


def test_cnnrnn_feature_mode(report):
    from tennis_amd import weights as W
    from tennis_amd.models.vision.definitions import CNNRNN
    for mode in ("gru", "lstm"):
        m = CNNRNN(None, num_classes=11, type=mode, hidden_size=128, prefix="cnnrnn0_")
        m.initialize()
        p = W.make_rnn_weights(3, mode, 1024, 128, f"cnnrnn0_{mode}0_")
        p.update(W.make_dense_weights(4, 11, 256, "cnnrnn0_dense0_"))
        m.set_params(p)
        x = np.abs(np.random.default_rng(0).normal(0, 1, (4, 9, 1024))).astype(np.float32) * 0.5
        got = m(x).cpu().numpy()
        ref, _ = vn.cnnrnn(x, p, mode, feats=True, rnn_prefix=f"cnnrnn0_{mode}0_")
        e = float(np.abs(got - ref).max())
        report[f"CNNRNN_{mode}_logits_maxabs_err"] = e
        assert got.shape == (4, 11) and e < 1e-4


def test_cnnrnn_on_frames(zoo, report):
    """TimeDistributed(backbone) -> bi-GRU -> max -> Dense on a (1,2,3,224,224) clip."""
    from tennis_amd import weights as W
    from tennis_amd.models.vision.definitions import CNNRNN
    fm, p = zoo
    m = CNNRNN(fm, num_classes=11, type="gru", hidden_size=128, prefix="cnnrnn1_")
    m.initialize()
    q = dict(p)
    q.update(W.make_rnn_weights(3, "gru", 1024, 128, "cnnrnn1_gru0_"))
    q.update(W.make_dense_weights(4, 11, 256, "cnnrnn1_dense0_"))
    m.set_params(q)
    clip = _frames(2, seed=77)[None]
    got = m(clip).cpu().numpy()
    ref, _ = vn.cnnrnn(clip, q, "gru", feats=False, rnn_prefix="cnnrnn1_gru0_", cls_prefix="cnnrnn1_dense0_")
    e = float(np.abs(got - ref).max())
    report["CNNRNN_frames_logits_maxabs_err"] = e
    assert e < TOL


def test_temporal_pooling_shares_classes(zoo):
    from tennis_amd.models.vision.definitions import TemporalPooling
    fm, p = zoo
    feats = np.abs(np.random.default_rng(2).normal(0, 1, (3, 5, 1024))).astype(np.float32)
    for pool in ("mean", "max"):
        tp = TemporalPooling(fm, pool=pool, num_classes=0, feats=True)   # evaluate.py:242-244
        assert tp.classes is fm.classes
        got = tp(feats).cpu().numpy()
        ref = vn.temporal_pooling(feats, p, pool=pool, feats=True)
        assert np.abs(got - ref).max() < 1e-4


def test_save_features_and_evaluate_model(zoo, tmp_path, report):
    """BASELINE config C1 at its full size: 32 frames of (3, 224, 224) through evaluate.py --save_feats, then evaluate."""
    from tennis_amd import evaluate as ev
    from tennis_amd.dataset import DataLoader, TennisSet
    from tennis_amd.metrics.vision import PRF1
    fm, p = zoo
    root = str(tmp_path / "data")
    ds = TennisSet(root=root, split="test", model_id="0006", save_feats=True, frames_per_video=16, data_shape=224)
    assert len(ds) == 32
    loader = DataLoader(ds, batch_size=32)
    assert ev.save_features(fm, loader, ds, verbose=False) == 32
    path = ds.save_feature_path(20)
    assert path == os.path.join(root, "features", "0006", "V007.mp4", "0000000000", "0000000004.npy")
    f = np.load(path)
    assert f.dtype == np.float32 and f.shape == (1024,)
    x = ds[20][0][None]
    ref = dn.densenet121_features(x.astype(np.float32), p)[0]       # the oracle on the UN-rounded fp32 frame the transform produces
    e = float(np.abs(f - ref).max())
    report["save_features_maxabs_err"] = e
    assert e < 1e-3          # (the encoder rounds the pixels to fp16 on the way in: inside the bar, measured 5.7e-4)
    assert ev.save_features(fm, loader, ds, verbose=False) == 0          # skip-if-exists (evaluate.py:318)

    ds2 = TennisSet(root=root, split="test", frames_per_video=16, data_shape=224)
    metric = PRF1(label_names=ds2.classes)
    results, gts = ev.evaluate_model(fm, DataLoader(ds2, batch_size=32), ds2, [metric])
    assert len(results) == 32 and all(v.shape == (11,) for v in results.values())
    k = ds2.get_image_path(ds2._frames_dir, "V006", 2)
    assert k in results and gts[k] == ds2.classes.index(ds2._samples[2][2])
    assert len(metric.get()) == 39 and metric.mat.sum() == 32

    # features on disk feed the temporal model through the same dataset contract (dataset.py:202-204)
    ds3 = TennisSet(root=root, split="test", window=3, feats_model="0006", frames_per_video=16, data_shape=224)
    xw, _, _ = ds3[1]
    assert xw.shape == (3, 1024)
    assert ev.main(["--root", root, "--model_id", "0007", "--save_feats", "--frames_per_video", "2",
                    "--batch_size", "4"]) == 0


def test_cnnrnn_full_size_c3(report):
    """BASELINE.json config C3 size: clip batch 32 x T=64 x F=1024 -> bi-GRU / bi-LSTM(128) -> max over T -> Dense(11),
    against the numpy oracle at the full size (the persistent recurrent kernel runs all 64 steps in one launch)."""
    from tennis_amd import weights as W
    from tennis_amd.models.vision.definitions import CNNRNN
    x = np.abs(np.random.default_rng(5).normal(0, 1, (32, 64, 1024))).astype(np.float32) * 0.5
    for mode in ("gru", "lstm"):
        m = CNNRNN(None, num_classes=11, type=mode, hidden_size=128, prefix="cnnrnn0_")
        m.initialize()
        p = W.make_rnn_weights(6, mode, 1024, 128, f"cnnrnn0_{mode}0_")
        p.update(W.make_dense_weights(7, 11, 256, "cnnrnn0_dense0_"))
        m.set_params(p)
        got = m(x).cpu().numpy()
        ref, _ = vn.cnnrnn(x, p, mode, feats=True, rnn_prefix=f"cnnrnn0_{mode}0_")
        e = float(np.abs(got - ref).max())
        report[f"CNNRNN_{mode}_C3_full_size_logits_maxabs_err"] = e
        assert got.shape == (32, 11) and e < 1e-4


@pytest.mark.parametrize("mode", ["gru", "lstm"])
def test_recurrence_h256_does_not_depend_on_the_kernel(mode):
    """H = 256: three clips run on ``rnn_recurrent_big_kernel`` (one row per workgroup; weights in registers + LDS + stream, h
    through DPP), the same clips inside a batch of 520 on the four-rows-per-workgroup kernel.  Both sum a gate's dot product over
    k in ascending order with one accumulator, so the outputs must be bit-identical (csrc/rnn_dot.h)."""
    from tennis_amd import weights as W
    from tennis_amd.engine import BiRNN
    rng = np.random.default_rng(3)
    F, H, T, B = 32, 256, 9, 520
    p = W.make_rnn_weights(5, mode, F, H, "r_")
    rnn = BiRNN(mode, F, H, p, "r_", max_rows=B * T)
    x = torch.from_numpy(rng.normal(0, 1, (B, T, F)).astype(np.float32)).cuda()
    vl = torch.from_numpy(rng.integers(1, T + 1, B).astype(np.int32)).cuda()
    big_seq, big_h, big_c = rnn(x, vl, return_state=True)
    seq, h, c = rnn(x[:3].contiguous(), vl[:3].contiguous(), return_state=True)
    assert torch.equal(seq, big_seq[:3]) and torch.equal(h, big_h[:, :3])
    if mode == "lstm":
        assert torch.equal(c, big_c[:, :3])
