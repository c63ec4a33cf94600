"""-m gpu: error behaviour of the C ABI (SURVEY §8b "Errors": integer codes + last-error string; the Python shim
raises RuntimeError) - shape / argument violations must fail loudly, never compute something else."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_encoder_argument_errors():
    from tennis_amd import _lib
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    enc = DenseNet121Features(p, 224, max_batch=2)
    x = torch.zeros((3, 3, 224, 224), dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="max_batch"):
        enc(x)                                                       # batch 3 > max_batch 2
    with pytest.raises(ValueError):
        enc(torch.zeros((1, 3, 100, 100), dtype=torch.float32, device="cuda"))   # wrong frame size for this encoder
    q = dict(p)
    del q["densenet0_stage3_conv7_weight"]
    with pytest.raises(RuntimeError, match="densenet0_stage3_conv7_weight"):
        DenseNet121Features(q, 224, max_batch=2)                     # a missing parameter is named
    q = dict(p)
    q["densenet0_conv0_weight"] = q["densenet0_conv0_weight"][:, :, :5]
    with pytest.raises(RuntimeError):
        DenseNet121Features(q, 224, max_batch=2)                     # wrong element count
    lib = _lib.load()
    assert lib.tn_densenet121_forward(None, None, 0, 1, None) != 0   # null handle: status code, no crash
    assert b"null" in lib.tn_last_error()


def test_rnn_and_head_argument_errors():
    from tennis_amd import weights as W
    from tennis_amd.engine import BiRNN, TemporalHeadTrainer
    p = W.make_rnn_weights(0, "gru", 32, 16, "r_")
    rnn = BiRNN("gru", 32, 16, p, "r_", max_rows=8)
    with pytest.raises(ValueError):
        rnn(torch.zeros((2, 3, 31), device="cuda"))                  # feature size mismatch
    with pytest.raises(RuntimeError):
        rnn(torch.zeros((3, 3, 32), device="cuda"))                  # 9 rows > max_rows 8
    with pytest.raises(RuntimeError):
        BiRNN("gru", 32, 1000, W.make_rnn_weights(0, "gru", 32, 1000, "r_"), "r_")   # 3*hidden > 1024
    hp = W.make_rnn_weights(1, "gru", 32, 16, "cnnrnn0_gru0_")
    hp.update(W.make_dense_weights(2, 11, 32, "cnnrnn0_dense0_"))
    tr = TemporalHeadTrainer(hp, 32, 16, 11, max_batch=4, max_steps=5)
    with pytest.raises(RuntimeError, match="exceed"):
        tr.forward_backward(torch.zeros((5, 5, 32), device="cuda"), torch.zeros(5, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError, match="unknown parameter"):
        tr.get("no_such_weight")
    del hp["cnnrnn0_dense0_bias"]
    with pytest.raises(RuntimeError, match="cnnrnn0_dense0_bias"):
        TemporalHeadTrainer(hp, 32, 16, 11)


def test_two_stream_model_is_refused():
    """Out of scope by design: the flow/two-stream model must say so instead of silently running."""
    from tennis_amd.models.vision.definitions import TwoStreamModel
    with pytest.raises(NotImplementedError):
        TwoStreamModel(None, None, 11)
