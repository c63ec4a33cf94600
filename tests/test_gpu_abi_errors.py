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


def test_gnmt_trainer_errors():
    """The captioner training handle: missing / mis-sized parameters, shapes beyond the handle, bad dropout rate and
    unknown names raise with the library's message; an LSTM model is refused by the driver."""
    import ctypes as C
    from tennis_amd import _lib, weights as W
    from tennis_amd.engine import GNMTTrainer
    p = W.make_gnmt_weights(0, "gru", 16, 8, 6, 12)
    tr = GNMTTrainer(p, 16, 8, 6, 12, max_batch=2, max_src_len=5, max_tgt_len=4)
    z = lambda *s: torch.zeros(s, dtype=torch.int32, device="cuda")
    src = torch.zeros((3, 5, 16), device="cuda")
    with pytest.raises(RuntimeError, match="exceed"):
        tr.forward_backward(src, z(3) + 5, z(3, 4), z(3) + 4)                     # batch 3 > 2
    with pytest.raises(RuntimeError, match="exceed"):
        tr.forward_backward(src[:2], z(2) + 5, z(2, 6), z(2) + 6)                # 6 target columns > 4
    with pytest.raises(RuntimeError, match="rate"):
        tr.set_dropout(1.0)
    with pytest.raises(KeyError):
        tr.get("gnmt_no_such_weight")
    lib = _lib.load()
    n, buf = C.c_int64(), (C.c_float * 4)()
    assert lib.tn_gnmt_trainer_read_param(tr.handle, b"gnmt_nope", 0, buf, 4, C.byref(n)) != 0
    assert b"unknown parameter" in lib.tn_last_error()
    assert lib.tn_gnmt_trainer_read_param(tr.handle, b"gnmt_tgt_proj_bias", 0, buf, 4, C.byref(n)) != 0     # 12 floats > 4
    assert b"too small" in lib.tn_last_error()
    q = dict(p)
    del q["gnmt_dec_attention_key_weight"]
    with pytest.raises(RuntimeError, match="dec_attention_key_weight"):
        GNMTTrainer(q, 16, 8, 6, 12)
    with pytest.raises(RuntimeError, match="wrong size"):
        GNMTTrainer(W.make_gnmt_weights(0, "lstm", 16, 8, 6, 12), 16, 8, 6, 12)   # 4-gate weights into the GRU trainer
    assert lib.tn_gnmt_trainer_destroy(None) == 0


def test_finetune_errors():
    import ctypes as C
    from tennis_amd import _lib, weights as W
    from tennis_amd.engine import FrameModelTrainer
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    with pytest.raises(RuntimeError, match="divisible by 32"):
        FrameModelTrainer(p, 200, 11, batch=2)
    tr = FrameModelTrainer(p, 64, 11, batch=2)                    # any side divisible by 32 works
    x = torch.zeros((3, 64, 64, 3), device="cuda")
    with pytest.raises(ValueError, match="FrameModelTrainer expects"):          # the host mirror checks batch and frame size ...
        tr.forward_backward(x, torch.zeros(3, dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError, match="FrameModelTrainer expects"):          # ... e.g. un-cropped frames (train.py transform)
        tr.forward_backward(torch.zeros((2, 96, 96, 3), device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"))
    lib0 = _lib.load()                                                           # ... and so does the C ABI underneath it
    lab = torch.zeros(3, dtype=torch.int32, device="cuda")
    assert lib0.tn_finetune_forward_backward(tr.handle, _lib.ptr(x), _lib.ptr(lab), 3, 64, 64, None, None) != 0
    assert b"batch must equal" in lib0.tn_last_error()
    assert lib0.tn_finetune_forward_backward(tr.handle, _lib.ptr(x), _lib.ptr(lab), 2, 96, 96, None, None) != 0
    assert b"frame size must equal" in lib0.tn_last_error()
    u8 = torch.randint(0, 256, (2, 64, 64, 3), device="cuda", dtype=torch.uint8)   # decoded frames: ToTensor + Normalize in the library
    from tennis_amd.engine import to_tensor_normalize
    ref = ((u8.float() / 255.0 - torch.tensor([0.485, 0.456, 0.406], device="cuda")) / torch.tensor([0.229, 0.224, 0.225], device="cuda"))
    assert float((to_tensor_normalize(u8) - ref).abs().max()) < 1e-6
    loss, logits = tr.forward_backward(torch.randn((2, 64, 64, 3), device="cuda"), torch.tensor([1, 5], dtype=torch.int32, device="cuda"))
    assert bool(torch.isfinite(loss).all()) and logits.shape == (2, 11)
    lib = _lib.load()
    n, buf = C.c_int64(), (C.c_float * 4)()
    assert lib.tn_finetune_read_param(tr.handle, b"densenet0_nope", 0, buf, 4, C.byref(n)) != 0
    assert b"unknown parameter" in lib.tn_last_error()
    q = dict(p)
    del q["densenet0_stage2_batchnorm5_running_var"]
    with pytest.raises(RuntimeError, match="stage2_batchnorm5_running_var"):
        FrameModelTrainer(q, 64, 11, batch=2)
    assert lib.tn_finetune_destroy(None) == 0


def test_encoder_create_ex_flags():
    """tn_densenet121_create_ex: unknown flags are errors; the exact-weights mode exists for every input size (round 6: the
    un-fused layer kernels and the 64 / 32 / 16 geometries of the tile kernel carry the hi + lo pass too)."""
    import ctypes as C
    from tennis_amd import _lib, weights as W
    ctx = _lib.default_context()
    arr, keep = _lib.make_params(W.make_densenet121_weights(0))
    h = C.c_void_p()
    assert ctx.lib.tn_densenet121_create_ex(ctx.handle, arr, len(arr), b"densenet0_", 224, 224, 2, 2, C.byref(h)) != 0
    assert b"unknown flag" in ctx.lib.tn_last_error()
    for size in (512, 236):
        assert ctx.lib.tn_densenet121_create_ex(ctx.handle, arr, len(arr), b"densenet0_", size, size, 2, _lib.ENC_EXACT_WEIGHTS, C.byref(h)) == 0
        assert ctx.lib.tn_densenet121_destroy(h) == 0
    assert ctx.lib.tn_densenet121_create_ex(ctx.handle, arr, len(arr), b"densenet0_", 224, 224, 2, _lib.ENC_EXACT_WEIGHTS, C.byref(h)) == 0
    assert ctx.lib.tn_densenet121_destroy(h) == 0


def test_jpeg_argument_errors():
    """tn_jpeg_*: null arguments and empty batches are status codes, a closed decoder raises, out= must fit"""
    from tennis_amd import _lib, image
    lib = _lib.load()
    ctx = _lib.default_context()
    h = C.c_void_p()
    assert lib.tn_jpeg_create(None, C.byref(h)) != 0 and b"null" in lib.tn_last_error()
    assert lib.tn_jpeg_create(ctx.handle, C.byref(h)) == 0
    assert lib.tn_jpeg_decode(h, None, None, 1, None, None, None) != 0 and b"null" in lib.tn_last_error()
    gold = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "jpeg_cases.npz"))
    data = gold["c420_q85__jpeg"].tobytes()
    ptrs = (C.c_void_p * 1)(C.cast(C.c_char_p(data), C.c_void_p))
    sizes = (C.c_size_t * 1)(len(data))
    out = torch.empty((1, 37, 53, 3), dtype=torch.uint8, device="cuda")
    assert lib.tn_jpeg_decode(h, ptrs, sizes, 0, _lib.ptr(out), None, None) != 0 and b"batch" in lib.tn_last_error()
    assert lib.tn_jpeg_decode(h, ptrs, sizes, 1, _lib.ptr(out), None, None) == 0
    assert np.array_equal(out[0].cpu().numpy(), gold["c420_q85__rgb"])
    assert lib.tn_jpeg_destroy(h) == 0 and lib.tn_jpeg_destroy(None) == 0
    assert lib.tn_jpeg_sync_passes(None) == 0
    dec = image.JpegDecoder()
    with pytest.raises(ValueError, match="out must be"):
        dec.decode([data], out=torch.empty((1, 10, 10, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="no files"):
        dec.decode([])
    dec.close()
    with pytest.raises(RuntimeError, match="closed"):
        dec.decode([data])


def test_round3_entry_point_errors():
    """The entry points added in round 3 refuse what they cannot do, with a message: calibration statistics on an
    exact-weights encoder or past max_batch, captioner layer counts the reference itself (or its attention cell) rejects,
    unknown flags, a 7x7 block the LDS-resident kernel has no room for, a communicator with a rank outside its world."""
    import ctypes as C
    from tennis_amd import _lib
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features, GNMTCaptioner
    p = W.make_densenet121_weights(0)
    enc = DenseNet121Features(p, 224, max_batch=2)
    with pytest.raises(RuntimeError, match="max_batch"):
        enc.input_means(torch.zeros((3, 224, 224, 3), dtype=torch.uint8, device="cuda"))
    ex = DenseNet121Features(W.make_densenet121_weights(0, fp16_model=False), 224, max_batch=2, exact_weights=True)
    with pytest.raises(RuntimeError, match="exact"):
        ex.input_means(torch.zeros((2, 224, 224, 3), dtype=torch.uint8, device="cuda"))
    # (round 6: the exact-weights mode exists for every input size - tests/test_gpu_parity_timed.py measures it at 236 / 448 / 512)
    assert DenseNet121Features(W.make_densenet121_weights(0, fp16_model=False), 512, max_batch=1, exact_weights=True).feature_dim == 4096
    g = W.make_gnmt_weights(1, "gru", 24, 16, 12, 30, num_layers=2, num_bi_layers=2)
    with pytest.raises(RuntimeError, match="num_bi_layers"):
        GNMTCaptioner(g, 24, 16, 12, 30, num_layers=2, num_bi_layers=2)
    with pytest.raises(RuntimeError, match="num_layers"):
        GNMTCaptioner(W.make_gnmt_weights(1, "gru", 24, 16, 12, 30, num_layers=1, num_bi_layers=0), 24, 16, 12, 30, num_layers=1, num_bi_layers=0)
    lib = _lib.load()
    ctx = _lib.default_context()
    h = C.c_void_p()
    arr, keep = _lib.make_params(W.make_gnmt_weights(1, "gru", 24, 16, 12, 30))
    assert lib.tn_gnmt_create_ex(ctx.handle, arr, len(arr), b"gnmt_", _lib.RNN_GRU, 24, 16, 12, 30, 2, 1, 4, 8, 2, 5, 64, C.byref(h)) != 0
    assert b"flag" in lib.tn_last_error()
    one = np.ones(128 * 64 + 128 * 96, np.float32)
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    assert lib.tn_dbg_block7_create(ctx.handle, 64, 2, vp(one), vp(one), vp(one), vp(one), vp(one), vp(np.ones(2 * 32 * 128 * 9, np.float32)), C.byref(h)) != 0
    assert b"unsupported" in lib.tn_last_error()
    hc = C.c_void_p()
    assert lib.tn_comm_create(ctx.handle, 3, 2, None, 0, C.byref(hc)) != 0      # rank 3 of 2
    assert lib.tn_last_error() != b""
