"""-m gpu: end-to-end fine-tuning step of the frame classifier (DenseNet-121 + Dense, BatchNorm in training mode, softmax CE,
SGD) through the C ABI vs oracle/densenet_train_torch.py (torch autograd on the CPU, float64)."""
import numpy as np
import pytest
import torch

from oracle import densenet_train_torch as dt
from oracle import train_np as tn

pytestmark = pytest.mark.gpu


def _setup(B, size=224, seed=5):
    from tennis_amd import weights as W
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    x = W.normalize_to_nchw_f32(W.synthetic_frames_u8(B, size, seed))
    y = np.random.default_rng(seed).integers(0, 11, B).astype(np.int32)
    return p, x, y


def _compare(tr, rg):
    """per parameter: max-abs error relative to the largest reference entry (floored at 1e-3 of the largest gradient entry
    of the whole model: gradients that are analytically ~0, e.g. a BatchNorm shift in front of another training-mode
    BatchNorm with every ReLU open, are pure rounding noise) and cosine similarity of the non-negligible ones"""
    floor = 1e-3 * max(np.abs(g).max() for g in rg.values())
    worst, worst_k, min_cos = 0.0, None, 1.0
    for k, g in rg.items():
        got = tr.get(k, gradient=True).astype(np.float64)
        err = np.abs(got - g).max() / max(floor, np.abs(g).max())
        if np.abs(g).max() > floor:
            min_cos = min(min_cos, float((got * g).sum() / max(1e-30, np.linalg.norm(got) * np.linalg.norm(g))))
        if err > worst:
            worst, worst_k = err, k
    return worst, worst_k, min_cos


def test_every_operator_backward_exact_with_open_relus(report):
    """All 364 gradients against autograd with the BatchNorm shifts raised by +4, so that (almost) no ReLU input lies near
    zero: float32 and float64 then take the same ReLU branches and every convolution / BatchNorm / pooling backward must agree
    to rounding.  (With the stock parameters a handful of the ~5 M ReLU inputs fall within float32 rounding of zero and
    take the other branch than in float64 — with 2 frames a single such element moves a 7x7-stage gradient by percents —
    which is a property of the comparison, not of either implementation; see the next test.)"""
    from tennis_amd.engine import FrameModelTrainer
    B = 2
    p, x, y = _setup(B)
    p = {k: (v + 4.0).astype(np.float32) if k.endswith("_beta") else v for k, v in p.items()}
    tr = FrameModelTrainer(p, 224, 11, batch=B)
    loss, logits = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    rl, rlog, rg, rstats = dt.loss_and_grads(p, x, y)
    assert np.abs(logits.cpu().numpy() - rlog).max() < 1e-3 * max(1.0, np.abs(rlog).max())
    worst, worst_k, min_cos = _compare(tr, rg)
    report["finetune_open_relu_grad_rel_err_worst"] = float(worst)
    assert worst < 2e-3 and min_cos > 0.999999, (worst_k, worst, min_cos)


def test_training_forward_backward_matches_autograd(report):
    from tennis_amd.engine import FrameModelTrainer
    B = 2
    p, x, y = _setup(B)
    tr = FrameModelTrainer(p, 224, 11, batch=B)
    loss, logits = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    rl, rlog, rg, rstats = dt.loss_and_grads(p, x, y)
    el = float(np.abs(logits.cpu().numpy() - rlog).max())
    report["finetune_logits_maxabs_err"] = el
    assert el < 1e-4 and np.abs(loss.cpu().numpy() - rl).max() < 1e-4, (el, loss.cpu().numpy(), rl)
    for bn in ("densenet0_batchnorm0", "densenet0_stage1_batchnorm1", "densenet0_stage3_batchnorm47", "densenet0_batchnorm4"):
        c = rstats[bn][0].shape[0]
        assert np.abs(tr.get(bn + "_batch_mean", shape=(c,)) - rstats[bn][0]).max() < 1e-4 * max(1.0, np.abs(rstats[bn][0]).max())
        assert np.abs(tr.get(bn + "_batch_var", shape=(c,)) - rstats[bn][1]).max() < 1e-4 * max(1.0, np.abs(rstats[bn][1]).max())
    # the classifier and the last layers' convolutions see no ReLU decision upstream of them: tight
    for k in ("framemodel0_dense0_weight", "framemodel0_dense0_bias", "densenet0_stage4_conv31_weight", "densenet0_stage4_conv30_weight"):
        g = rg[k]
        assert np.abs(tr.get(k, gradient=True) - g).max() < 1e-4 * np.abs(g).max(), k
    worst, worst_k, min_cos = _compare(tr, rg)
    report["finetune_grad_rel_err_worst"] = float(worst)
    report["finetune_grad_min_cosine"] = float(min_cos)
    assert min_cos > 0.995 and worst < 0.3, (worst_k, worst, min_cos)     # a few float32 / float64 ReLU branch differences
    # running statistics: 0.9 * old + 0.1 * batch
    bn = "densenet0_stage2_batchnorm3"
    exp = 0.9 * p[bn + "_running_mean"] + 0.1 * rstats[bn][0]
    assert np.abs(tr.get(bn + "_running_mean") - exp).max() < 1e-4


def test_sgd_steps_reduce_the_loss():
    """A few steps of train.py's recipe (momentum 0.9, wd 1e-4, rescale 1/batch) on one fixed batch; the first update is
    oracle/train_np.py::sgd_momentum applied to the library's own gradients."""
    from tennis_amd.engine import FrameModelTrainer
    B = 4
    p, x, y = _setup(B, seed=9)
    tr = FrameModelTrainer(p, 224, 11, batch=B)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    loss, _ = tr.forward_backward(xd, yd)
    first = float(loss.mean())
    names = ["framemodel0_dense0_weight", "densenet0_stage2_conv5_weight", "densenet0_conv0_weight", "densenet0_stage4_batchnorm7_gamma"]
    g0 = {k: tr.get(k, gradient=True) for k in names}
    tr.step(B, 0.01, 0.9, 1e-4)
    p1, _ = tn.sgd_momentum({k: p[k].astype(np.float64) for k in names}, g0, {}, 0.01, 0.9, 1e-4, 1.0 / B)
    for k in names:
        assert np.abs(tr.get(k) - p1[k]).max() < 1e-6 * max(1.0, np.abs(p1[k]).max()), k
    for _ in range(12):
        loss, _ = tr.forward_backward(xd, yd)
        tr.step(B, 0.01, 0.9, 1e-4)
    assert float(loss.mean()) < 0.7 * first and bool(torch.isfinite(tr.grads).all())


def test_train_model_driver_with_the_frame_classifier(tmp_path):
    """tennis_amd.train.train_model (reference train.py:388-499) drives the end-to-end classifier exactly as it drives the
    temporal head: LR schedule, metric updates, per-epoch parameter files that load back into a FrameModel."""
    from tennis_amd.engine import FrameModelTrainer
    from tennis_amd.metrics.vision import PRF1
    from tennis_amd.model_zoo import get_model
    from tennis_amd.models.vision.definitions import FrameModel
    from tennis_amd.train import Trainer, train_model
    B = 4
    p, x, y = _setup(B, seed=11)
    head = FrameModelTrainer(p, 224, 11, batch=B)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    tr = Trainer(head, "sgd", {"learning_rate": 0.01, "momentum": 0.9, "wd": 1e-4})
    metric = PRF1(label_names=[str(i) for i in range(11)])
    mirror = FrameModel(get_model("DenseNet121", pretrained=True, seed=5).features, 11, prefix="framemodel0_")
    mirror.initialize()
    mirror.classes._materialize(1024)
    hist = train_model(head, lambda: [(xd, yd)] * 3, [metric], tr, epochs=3, batch_size=B, lr_steps=(1, 2), lr_factor=0.75,
                       save_dir=str(tmp_path), log=lambda *_: None, model=mirror)
    assert len(hist) == 3 and hist[-1]["loss"] < hist[0]["loss"] and abs(hist[-1]["lr"] - 0.01 * 0.75 ** 2) < 1e-9
    from tennis_amd.params_io import is_mxnet_params
    assert is_mxnet_params(str(tmp_path / "0002.params"))    # train.py:497: the MXNet container with Gluon's structural names
    fm = FrameModel(get_model("DenseNet121", pretrained=True, seed=3).features, 11, prefix="framemodel0_")
    fm.initialize()
    fm.classes._materialize(1024)
    fm.load_parameters(str(tmp_path / "0002.params"))
    st = head.state_dict()
    for k in ("densenet0_stage3_conv7_weight", "densenet0_batchnorm2_running_var", "framemodel0_dense0_bias"):
        got = fm.collect_params()[k].data
        ref = st[k].astype(np.float16).astype(np.float32) if k.endswith("conv7_weight") else st[k]   # the served model rounds conv weights to fp16
        assert np.allclose(got, ref, atol=1e-6), k
    logits = fm(xd).cpu().numpy()                            # inference with the running statistics just trained
    assert logits.shape == (B, 11) and np.isfinite(logits).all()
