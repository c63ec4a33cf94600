"""-m gpu: training step of the temporal head (bi-GRU | bi-LSTM -> max over T -> Dense -> softmax CE, SGD momentum + wd)
through the C ABI vs oracle/train_np.py (itself pinned to torch autograd on the CPU)."""
import numpy as np
import pytest
import torch

from oracle import train_np as tn

pytestmark = pytest.mark.gpu


def _setup(seed, B, T, F, H, C_, cell="gru"):
    from tennis_amd import weights as W
    p = W.make_rnn_weights(seed, cell, F, H, f"cnnrnn0_{cell}0_")
    p.update(W.make_dense_weights(seed + 1, C_, 2 * H, "cnnrnn0_dense0_"))
    rng = np.random.default_rng(seed)
    x = (np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)
    y = rng.integers(0, C_, B).astype(np.int32)
    return p, x, y


@pytest.mark.parametrize("cell", ["gru", "lstm"])
@pytest.mark.parametrize("B,T,F,H", [(3, 5, 24, 8), (6, 9, 64, 32), (32, 64, 1024, 128)])   # last: BASELINE config C3
def test_gradients_and_sgd_step(report, B, T, F, H, cell):
    from tennis_amd.engine import TemporalHeadTrainer
    C_ = 11
    p, x, y = _setup(4, B, T, F, H, C_, cell)
    tr = TemporalHeadTrainer(p, F, H, C_, max_batch=B, max_steps=T, type=cell)
    loss, logits = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    rl, rlg, rg = tn.forward_backward(x, y, p, cell=cell)
    assert np.abs(loss.cpu().numpy() - rl).max() < 1e-4 and np.abs(logits.cpu().numpy() - rlg).max() < 1e-4
    worst = 0.0
    for k, g in rg.items():
        got = tr.get(k, gradient=True).reshape(g.shape)
        err = np.abs(got - g).max() / max(1e-6, np.abs(g).max())
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    report[f"train_head_{cell}_grad_rel_err_B{B}_T{T}_F{F}"] = float(worst)
    # one SGD step with Gluon's rescale 1/batch_size, momentum 0.9, wd 1e-4 (train.py flags), then a second one
    lr, mo, wd = 1e-2, 0.9, 1e-4
    tr.step(B, lr, mo, wd)
    p1, m1 = tn.sgd_momentum({k: v.astype(np.float64) for k, v in p.items()}, rg, {}, lr, mo, wd, 1.0 / B)
    st = tr.state_dict()
    for k in p1:
        assert np.abs(st[k] - p1[k]).max() < 1e-5 * max(1.0, np.abs(p1[k]).max()), k
    tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    _, _, rg2 = tn.forward_backward(x, y, {k: v.astype(np.float32) for k, v in p1.items()}, cell=cell)
    tr.step(B, lr, mo, wd)
    p2, _ = tn.sgd_momentum(p1, rg2, m1, lr, mo, wd, 1.0 / B)
    st = tr.state_dict()
    for k in p2:
        assert np.abs(st[k] - p2[k]).max() < 5e-5 * max(1.0, np.abs(p2[k]).max()), k


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_training_reduces_the_loss_and_grads_view(cell):
    """A hundred steps on one fixed batch drive the summed loss down; the flat gradient view is what a
    data-parallel all-reduce would operate on."""
    from tennis_amd.engine import TemporalHeadTrainer
    B, T, F, H, C_ = 16, 12, 48, 16, 11
    p, x, y = _setup(7, B, T, F, H, C_, cell)
    tr = TemporalHeadTrainer(p, F, H, C_, max_batch=B, max_steps=T, type=cell)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    first = None
    for i in range(120):     # (the fp64 oracle's curve: GRU 2.41 -> 0.05, LSTM 2.41 -> 0.82)
        loss, _ = tr.forward_backward(xd, yd)
        if first is None:
            first = float(loss.mean())
            g = tr.grads
            assert g.shape == (tr.numel,) and torch.isfinite(g).all() and float(g.abs().max()) > 0
        tr.step(B, 0.05, 0.9, 1e-4)
    assert float(loss.mean()) < 0.7 * first


def test_train_model_driver(tmp_path):
    """tennis_amd.train.train_model (reference train.py:388-499): LR schedule, metric updates, per-epoch parameter
    files; the loss falls over the epochs on a separable synthetic task."""
    from tennis_amd.engine import TemporalHeadTrainer
    from tennis_amd.metrics.vision import PRF1
    from tennis_amd.train import Trainer, train_model
    B, T, F, H, C_ = 16, 8, 32, 16, 11
    p, _, _ = _setup(9, B, T, F, H, C_)
    rng = np.random.default_rng(3)
    protos = rng.normal(0, 1, (C_, F)).astype(np.float32)

    def batches():
        for _ in range(6):
            y = rng.integers(0, C_, B)
            x = (protos[y][:, None, :] + 0.3 * rng.normal(0, 1, (B, T, F))).astype(np.float32)
            yield torch.from_numpy(x).cuda(), torch.from_numpy(y.astype(np.int32)).cuda()

    head = TemporalHeadTrainer(p, F, H, C_, max_batch=B, max_steps=T)
    trainer = Trainer(head, "sgd", {"learning_rate": 0.1, "momentum": 0.9, "wd": 1e-4})
    metric = PRF1(label_names=[str(i) for i in range(C_)])
    hist = train_model(head, batches, [metric], trainer, epochs=5, batch_size=B, lr_steps=(2, 4), lr_factor=0.5,
                       save_dir=str(tmp_path), log=lambda s: None)
    assert [round(h["lr"], 6) for h in hist] == [0.1, 0.1, 0.05, 0.05, 0.025]
    assert hist[-1]["loss"] < 0.5 * hist[0]["loss"]
    assert metric.mat.sum() == 6 * B                                  # the last epoch's samples
    z = np.load(str(tmp_path / "0004.npz"))          # no mirror model handed over: prefixed names in an .npz
    assert set(z.files) == set(p) and z["cnnrnn0_dense0_weight"].shape == (C_, 2 * H)


def test_train_main_pipeline(tmp_path, capsys):
    """The reference's workflow for the temporal model end to end on the synthetic source: evaluate --save_feats writes the
    per-frame .npy features, train --feats_model --window --temp_pool gru trains the head on windows of them, and the saved
    parameters load back; then one epoch of the end-to-end frame classifier (train --window 1)."""
    from tennis_amd import evaluate as ev, train as tr
    root = str(tmp_path / "data")
    common = ["--root", root, "--frames_per_video", "12", "--data_shape", "224", "--model_id", "0001"]
    for split in ("train", "val"):
        assert ev.main(common + ["--split", split, "--save_feats", "--batch_size", "8"]) == 0
    exp = str(tmp_path / "exp")
    args = ["--root", root, "--frames_per_video", "12", "--data_shape", "224", "--model_id", "0002", "--feats_model", "0001", "--window", "4",
            "--temp_pool", "gru", "--epochs", "3", "--batch_size", "8", "--lr", "0.01", "--lr_steps", "1, 2", "--exp_root", exp]
    assert tr.main(args) == 0
    out = capsys.readouterr().out
    assert "[Finished] best epoch" in out and (tmp_path / "exp" / "0002" / "0002.params").exists()
    # train.py:487-489,497: one "epoch<TAB>AVG_NB_f1" line per epoch, NNNN.params in the MXNet container with Gluon's
    # structural names (what the reference's load_parameters reads)
    from tennis_amd.params_io import is_mxnet_params, load_mxnet_params
    from tennis_amd.train import best_epoch_from_scores
    lines = (tmp_path / "exp" / "0002" / "scores.txt").read_text().splitlines()
    assert [ln.split("\t")[0] for ln in lines] == ["0", "1", "2"]
    f2 = str(tmp_path / "exp" / "0002" / "0002.params")
    assert is_mxnet_params(f2) and "rnn.l0_i2h_weight" in load_mxnet_params(f2) and "classes.weight" in load_mxnet_params(f2)
    best_ep, best_sc = best_epoch_from_scores(str(tmp_path / "exp" / "0002" / "scores.txt"))
    # evaluate.py:223-240: testing loads the best epoch of scores.txt
    assert ev.main(["--root", root, "--frames_per_video", "12", "--data_shape", "224", "--model_id", "0002", "--feats_model", "0001",
                    "--window", "4", "--temp_pool", "gru", "--split", "val", "--batch_size", "8", "--exp_root", exp]) == 0
    out = capsys.readouterr().out
    assert "Testing best model from Epoch %d" % best_ep in out and "%04d.params" % best_ep in out
    assert "Test_AVG_NB_f1={:.3f}".format(best_sc) in out          # the validation score of that epoch, reproduced
    # train.py:286-295: a second run resumes after the newest NNNN.params
    assert tr.main(args[:args.index("--epochs")] + ["--epochs", "4"] + args[args.index("--epochs") + 2:]) == 0
    out = capsys.readouterr().out
    assert "Loaded model params" in out and "0002.params" in out and (tmp_path / "exp" / "0002" / "0003.params").exists()
    assert len((tmp_path / "exp" / "0002" / "scores.txt").read_text().splitlines()) == 4
    # end-to-end frame classifier, one epoch
    args = ["--root", root, "--frames_per_video", "8", "--data_shape", "224", "--model_id", "0003", "--window", "1", "--epochs", "1",
            "--batch_size", "4", "--exp_root", exp]
    assert tr.main(args) == 0
    assert (tmp_path / "exp" / "0003" / "0000.params").exists()
