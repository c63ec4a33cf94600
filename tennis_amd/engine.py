"""Device-side building blocks: thin Python handles over the C ABI.

Each class owns one opaque library handle (weights + workspace live in HBM
inside it) and launches on the Context's HIP stream.  Tensors crossing this
boundary are torch CUDA tensors used purely as device buffers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr


def _on_ctx_device(ctx, x: torch.Tensor, what: str):
    """A handle's weights and workspace live on its context's GPU: inputs must be there too (one process per GPU)."""
    if not x.is_cuda:
        raise ValueError(f"{what}: input must already be on the GPU")
    if x.device.index != ctx.device:
        raise ValueError(f"{what}: input is on cuda:{x.device.index} but this handle was built on cuda:{ctx.device}")


def _layout_of(x: torch.Tensor, size_hw):
    """Pick the tn_layout of a frame batch from dtype/shape (reference frames are
    NCHW float32 after ToTensor+Normalize, evaluate.py:96-97)."""
    h, w = size_hw
    if x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and tuple(x.shape[2:]) == (h, w):
        return _lib.LAYOUT_NCHW_F32
    if x.dtype == torch.float16 and x.dim() == 4 and x.shape[3] == 3 and tuple(x.shape[1:3]) == (h, w):
        return _lib.LAYOUT_NHWC_F16
    if x.dtype == torch.uint8 and x.dim() == 4 and x.shape[3] == 3 and tuple(x.shape[1:3]) == (h, w):
        return _lib.LAYOUT_NHWC_U8
    raise ValueError(f"unsupported frame batch: shape {tuple(x.shape)} dtype {x.dtype} for a {h}x{w} encoder "
                     "(expected NCHW float32, NHWC float16 or NHWC uint8)")


class DenseNet121Features:
    """``get_model('DenseNet121').features`` on the GPU (reference evaluate.py:125)."""

    def __init__(self, params: dict, size: int | tuple = 224, max_batch: int = 256, prefix: str = "densenet0_",
                 ctx: _lib.Context | None = None, exact_weights: bool = False):
        """``exact_weights``: keep the fp32 convolution weights of the dense layers / transitions as hi + lo fp16
        pairs (TN_ENC_EXACT_WEIGHTS) instead of rounding them to fp16 once — for parameters that were NOT converted
        with ``weights.as_fp16_model`` (a trained fp32 checkpoint) and a 1e-3 agreement with their fp32 evaluation."""
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.size = (size, size) if isinstance(size, int) else tuple(size)
        self.max_batch = max_batch
        arr, keep = _lib.make_params({k: v for k, v in params.items() if k.startswith(prefix)})
        h = C.c_void_p()
        self.exact_weights = bool(exact_weights)
        check(self.lib.tn_densenet121_create_ex(self.ctx.handle, arr, len(arr), prefix.encode(), self.size[0], self.size[1],
                                                max_batch, _lib.ENC_EXACT_WEIGHTS if exact_weights else 0, C.byref(h)),
              "tn_densenet121_create")
        del keep
        self.handle = h
        self.feature_dim = self.lib.tn_densenet121_feature_dim(h)
        self.workspace_bytes = self.lib.tn_densenet121_workspace_bytes(h)

    def __call__(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        _on_ctx_device(self.ctx, x, "DenseNet121Features")
        x = x.contiguous()
        layout = _layout_of(x, self.size)
        b = x.shape[0]
        if out is None:
            out = torch.empty((b, self.feature_dim), dtype=torch.float32, device=x.device)
        check(self.lib.tn_densenet121_forward(self.handle, ptr(x), layout, b, ptr(out)), "tn_densenet121_forward")
        return out

    def set_pipelined(self, on: bool = True):
        """Consecutive calls overlap on the library's side streams; the caller orders the results with ``join`` (see
        ``tn_densenet121_set_pipelined`` in include/tennis_hip.h)."""
        check(self.lib.tn_densenet121_set_pipelined(self.handle, 1 if on else 0), "tn_densenet121_set_pipelined")

    def join(self, lag: int = 0):
        """The context's stream waits for the last call (``lag=0``) or the one before it (``lag=1``)."""
        check(self.lib.tn_densenet121_join(self.handle, lag), "tn_densenet121_join")

    def profile(self, x: torch.Tensor):
        """One forward with every launch bracketed by HIP events -> list of dicts."""
        x = x.contiguous()
        layout = _layout_of(x, self.size)
        b = x.shape[0]
        out = torch.empty((b, self.feature_dim), dtype=torch.float32, device=x.device)
        stats = (_lib.TnKernelStat * 16)()
        n = C.c_int(0)
        check(self.lib.tn_densenet121_profile(self.handle, ptr(x), layout, b, ptr(out), stats, 16, C.byref(n)),
              "tn_densenet121_profile")
        return [dict(name=stats[i].name.decode(), launches=stats[i].launches, ms=stats[i].ms,
                     flops=stats[i].flops, bytes=stats[i].bytes) for i in range(n.value)], out

    def input_means(self, x: torch.Tensor, prefix: str = "densenet0_") -> dict:
        """Calibration statistics (``tn_densenet121_input_means``): ``{conv weight name: mean of every input channel of that
        convolution}`` over the frames ``x``, for the 119 convolutions behind the stem - what
        ``weights.as_fp16_model(params, input_means=...)`` needs."""
        from . import weights as W
        x = x.contiguous()
        layout = _layout_of(x, self.size)
        convs, _, _ = W.densenet121_layout()
        convs = [c for c in convs if c["kind"] != "stem"]
        buf = np.empty(sum(c["cin"] for c in convs), dtype=np.float32)
        n = C.c_int64(0)
        check(self.lib.tn_densenet121_input_means(self.handle, ptr(x), layout, x.shape[0], buf.ctypes.data_as(C.c_void_p), buf.size,
                                                  C.byref(n)), "tn_densenet121_input_means")
        assert n.value == buf.size
        out, o = {}, 0
        for c in convs:
            out[prefix + c["name"] + "_weight"] = buf[o:o + c["cin"]].copy()
            o += c["cin"]
        return out

    def read_tap(self, tap: str, batch: int) -> np.ndarray:
        buf = np.empty(self._tap_numel(batch), dtype=np.float32)
        n = C.c_size_t(0)
        check(self.lib.tn_densenet121_read_tap(self.handle, tap.encode(), batch,
                                               buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n)),
              "tn_densenet121_read_tap")
        return buf[:n.value]

    def _tap_numel(self, batch):
        hs = (self.size[0] - 1) // 2 + 1
        ws = (self.size[1] - 1) // 2 + 1
        return batch * hs * ws * 64  # stem / stage1 taps are the largest

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_densenet121_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Dense:
    """``nn.Dense(units, flatten=True)`` (reference definitions.py:25)."""

    def __init__(self, weight: np.ndarray, bias: np.ndarray | None, ctx: _lib.Context | None = None):
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        w = np.ascontiguousarray(weight, dtype=np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
        self.units, self.in_units = w.shape
        h = C.c_void_p()
        check(self.lib.tn_dense_create(self.ctx.handle, w.ctypes.data_as(C.c_void_p),
                                       None if b is None else b.ctypes.data_as(C.c_void_p),
                                       self.units, self.in_units, C.byref(h)), "tn_dense_create")
        self.handle = h

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _on_ctx_device(self.ctx, x, "Dense")
        x = x.reshape(x.shape[0], -1).contiguous().float()
        if x.shape[1] != self.in_units:
            raise ValueError(f"Dense expects {self.in_units} input units, got {x.shape[1]}")
        y = torch.empty((x.shape[0], self.units), dtype=torch.float32, device=x.device)
        check(self.lib.tn_dense_forward(self.handle, ptr(x), x.shape[0], ptr(y)), "tn_dense_forward")
        return y

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_dense_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class BiRNN:
    """``mx.gluon.rnn.GRU/LSTM(hidden, layout='NTC', bidirectional=...)`` (definitions.py:94-96)."""

    def __init__(self, mode: str, input_size: int, hidden: int, params: dict, prefix: str,
                 bidirectional: bool = True, max_rows: int = 4096, ctx: _lib.Context | None = None):
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.mode, self.hidden, self.input_size = mode, hidden, input_size
        self.dirs = 2 if bidirectional else 1
        self.max_rows = max_rows
        arr, keep = _lib.make_params({k: v for k, v in params.items() if k.startswith(prefix)})
        h = C.c_void_p()
        kind = _lib.RNN_GRU if mode == "gru" else _lib.RNN_LSTM
        check(self.lib.tn_birnn_create(self.ctx.handle, kind, input_size, hidden, arr, len(arr), prefix.encode(),
                                       1 if bidirectional else 0, max_rows, C.byref(h)), "tn_birnn_create")
        del keep
        self.handle = h

    def __call__(self, x: torch.Tensor, valid_length: torch.Tensor | None = None, return_state: bool = False):
        _on_ctx_device(self.ctx, x, "BiRNN")
        x = x.contiguous().float()
        b, t, f = x.shape
        if f != self.input_size:
            raise ValueError(f"rnn expects {self.input_size} input features, got {f}")
        seq = torch.empty((b, t, self.dirs * self.hidden), dtype=torch.float32, device=x.device)
        hl = torch.empty((self.dirs, b, self.hidden), dtype=torch.float32, device=x.device)
        cl = torch.zeros_like(hl)
        vl = None if valid_length is None else valid_length.to(device=x.device, dtype=torch.int32).contiguous()
        check(self.lib.tn_birnn_forward(self.handle, ptr(x), b, t, ptr(vl), ptr(seq), ptr(hl), ptr(cl)),
              "tn_birnn_forward")
        return (seq, hl, cl) if return_state else seq

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_birnn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def temporal_pool(x: torch.Tensor, kind: str, ctx: _lib.Context | None = None) -> torch.Tensor:
    """``F.max(x, axis=1)`` / ``F.mean(x, axis=1)`` (definitions.py:66-69,107)."""
    ctx = ctx or _lib.default_context(x.device.index)
    x = x.contiguous().float()
    b, t = x.shape[:2]
    f = int(np.prod(x.shape[2:]))
    y = torch.empty((b,) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    check(ctx.lib.tn_temporal_pool(ctx.handle, ptr(x), b, t, f, _lib.POOL_MEAN if kind == "mean" else _lib.POOL_MAX,
                                   ptr(y)), "tn_temporal_pool")
    return y


def to_tensor_normalize(x: torch.Tensor, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                        ctx: _lib.Context | None = None) -> torch.Tensor:
    """``transforms.ToTensor()`` + ``transforms.Normalize(mean, std)`` (reference evaluate.py:96-97) on a uint8
    (..., H, W, 3) device batch -> float32 of the same (NHWC) shape."""
    ctx = ctx or _lib.default_context(x.device.index)
    _on_ctx_device(ctx, x, "to_tensor_normalize")
    if x.dtype != torch.uint8 or x.shape[-1] != 3:
        raise ValueError(f"expected uint8 (..., H, W, 3) frames, got {tuple(x.shape)} {x.dtype}")
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    check(ctx.lib.tn_to_tensor_normalize(ctx.handle, ptr(x), x.numel() // 3, m, sd, ptr(y)), "tn_to_tensor_normalize")
    return y


class GNMTCaptioner:
    """Encoder + attention decoder + beam search of the reference captioner on the GPU
    (train_gnmt.py:223-252 assembly; evaluate() at train_gnmt.py:264-302)."""

    def __init__(self, params: dict, input_size: int, hidden: int, embed: int, vocab: int, beam: int = 4,
                 max_length: int = 150, max_batch: int = 32, max_src_len: int = 640, prefix: str = "gnmt_",
                 cell_type: str = "gru", num_layers: int = 2, num_bi_layers: int = 1, use_residual: bool = False,
                 ctx: _lib.Context | None = None):
        """``num_layers`` / ``num_bi_layers`` / ``use_residual`` as in ``get_gnmt_encoder_decoder`` (reference gnmt.py:407-455):
        2 <= num_layers, num_bi_layers < num_layers (gnmt.py:78-80 and the attention's key width, see tn_gnmt_create_ex)."""
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.hidden, self.beam, self.max_length, self.vocab = hidden, beam, max_length, vocab
        arr, keep = _lib.make_params({k: v for k, v in params.items() if k.startswith(prefix)})
        h = C.c_void_p()
        kind = _lib.RNN_GRU if cell_type == "gru" else _lib.RNN_LSTM
        check(self.lib.tn_gnmt_create_ex(self.ctx.handle, arr, len(arr), prefix.encode(), kind, input_size, hidden, embed,
                                         vocab, num_layers, num_bi_layers, max_batch, max_src_len, beam, max_length,
                                         1 if use_residual else 0, C.byref(h)), "tn_gnmt_create")
        del keep
        self.handle = h
        self._batch = 0

    def encode(self, src: torch.Tensor, valid_length: torch.Tensor) -> torch.Tensor:
        src = src.contiguous().float()
        b, t, _ = src.shape
        vl = valid_length.to(device=src.device).round().to(torch.int32).contiguous()
        mem = torch.empty((b, t, self.hidden), dtype=torch.float32, device=src.device)
        check(self.lib.tn_gnmt_encode(self.handle, ptr(src), ptr(vl), b, t, ptr(mem)), "tn_gnmt_encode")
        self._batch = b
        return mem

    def decode_seq(self, tgt: torch.Tensor) -> torch.Tensor:
        """Teacher-forced logits (B, L, V) for target tokens (B, L) after encode()."""
        tgt = tgt.to(device=torch.device("cuda", self.ctx.device)).round().to(torch.int32).contiguous()
        b, l = tgt.shape
        logits = torch.empty((b, l, self.vocab), dtype=torch.float32, device=tgt.device)
        check(self.lib.tn_gnmt_decode_seq(self.handle, ptr(tgt), l, l, ptr(logits)), "tn_gnmt_decode_seq")
        return logits

    def beam_search(self, bos: int, eos: int, alpha: float = 1.0, K: float = 5.0, max_length: int | None = None):
        ml = self.max_length if max_length is None else max_length
        b, dev = self._batch, torch.device("cuda", self.ctx.device)
        samples = torch.empty((b, self.beam, self.max_length + 2), dtype=torch.int32, device=dev)
        scores = torch.empty((b, self.beam), dtype=torch.float32, device=dev)
        vlen = torch.empty((b, self.beam), dtype=torch.int32, device=dev)
        n = C.c_int(0)
        check(self.lib.tn_gnmt_beam_search(self.handle, bos, eos, alpha, K, ml, ptr(samples), ptr(scores), ptr(vlen),
                                           C.byref(n)), "tn_gnmt_beam_search")
        return samples[:, :, :n.value].contiguous(), scores, vlen

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_gnmt_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def masked_softmax_ce(logits: torch.Tensor, labels: torch.Tensor, valid_length: torch.Tensor,
                      ctx: _lib.Context | None = None) -> torch.Tensor:
    """``gluonnlp.loss.MaskedSoftmaxCELoss`` (reference train_gnmt.py:256,281): (B,) losses."""
    ctx = ctx or _lib.default_context(logits.device.index)
    logits = logits.contiguous().float()
    b, l, v = logits.shape
    lab = labels.to(logits.device).round().to(torch.int32).contiguous()
    vl = valid_length.to(logits.device).round().to(torch.int32).contiguous()
    loss = torch.empty((b,), dtype=torch.float32, device=logits.device)
    check(ctx.lib.tn_masked_softmax_ce(ctx.handle, ptr(logits), ptr(lab), lab.shape[1], ptr(vl), b, l, v, ptr(loss)),
          "tn_masked_softmax_ce")
    return loss


class TemporalHeadTrainer:
    """Training step of ``CNNRNN(model=None, type='gru' | 'lstm')`` in feature mode (reference definitions.py:94-110) the way
    train.py drives it: ``SoftmaxCrossEntropyLoss`` per sample (:324), ``ag.backward`` of the per-sample losses and
    ``gluon.Trainer(params, 'sgd', {learning_rate, momentum, wd}).step(batch_size)`` (:298-299, :410-424).

    ``forward_backward`` leaves the gradient of the SUM of the per-sample losses in a flat device buffer
    (``grads``); with several ranks all-reduce that buffer (``torch.distributed.all_reduce(trainer.grads)``)
    before ``step(batch_size)``, whose ``rescale_grad = 1 / batch_size`` is Gluon's."""

    def __init__(self, params: dict, input_size: int, hidden: int = 128, classes: int = 11, max_batch: int = 32,
                 max_steps: int = 64, rnn_prefix: str | None = None, dense_prefix: str = "cnnrnn0_dense0_",
                 ctx: _lib.Context | None = None, type: str = "gru"):
        if type not in ("gru", "lstm"):
            raise ValueError(f"type must be 'gru' or 'lstm', got {type!r}")
        if rnn_prefix is None:
            rnn_prefix = f"cnnrnn0_{type}0_"
        self.type, self.gates = type, 3 if type == "gru" else 4
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.input_size, self.hidden, self.classes = input_size, hidden, classes
        self.rnn_prefix, self.dense_prefix = rnn_prefix, dense_prefix
        arr, keep = _lib.make_params({k: v for k, v in params.items() if k.startswith(rnn_prefix) or k.startswith(dense_prefix)})
        h = C.c_void_p()
        check(self.lib.tn_head_create(self.ctx.handle, _lib.RNN_GRU if type == "gru" else _lib.RNN_LSTM, input_size, hidden, classes, arr, len(arr), rnn_prefix.encode(),
                                      dense_prefix.encode(), max_batch, max_steps, C.byref(h)), "tn_head_create")
        del keep
        self.handle = h
        pw, pg, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.tn_head_buffers(h, C.byref(pw), C.byref(pg), C.byref(n)), "tn_head_buffers")
        self.numel = n.value
        self._pw, self._pg = pw.value, pg.value

    def _view(self, addr):
        """torch view of a flat fp32 device buffer owned by the library (for all-reduce / inspection)."""
        class _Arr:
            __cuda_array_interface__ = {"shape": (self.numel,), "typestr": "<f4", "data": (addr, False), "version": 3}
        return torch.as_tensor(_Arr(), device=f"cuda:{self.ctx.device}")

    @property
    def grads(self) -> torch.Tensor:
        return self._view(self._pg)

    @property
    def params(self) -> torch.Tensor:
        return self._view(self._pw)

    def forward_backward(self, x: torch.Tensor, labels: torch.Tensor):
        x = x.contiguous().float()
        b, t, f = x.shape
        labels = labels.to(device=x.device, dtype=torch.int32).contiguous()
        loss = torch.empty((b,), dtype=torch.float32, device=x.device)
        logits = torch.empty((b, self.classes), dtype=torch.float32, device=x.device)
        check(self.lib.tn_head_forward_backward(self.handle, ptr(x), ptr(labels), b, t, ptr(loss), ptr(logits)),
              "tn_head_forward_backward")
        return loss, logits

    def step(self, batch_size: int, lr: float, momentum: float = 0.9, wd: float = 1e-4):
        check(self.lib.tn_head_sgd_step(self.handle, lr, momentum, wd, 1.0 / batch_size), "tn_head_sgd_step")

    def get(self, name: str, gradient: bool = False) -> np.ndarray:
        cap = self.gates * self.hidden * max(self.input_size, self.hidden, 2 * self.classes) + 16
        out = np.empty(cap, np.float32)
        n = C.c_int64()
        check(self.lib.tn_head_read_param(self.handle, name.encode(), 1 if gradient else 0,
                                          out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n)), "tn_head_read_param")
        return out[:n.value].copy()

    def state_dict(self) -> dict:
        h, f, c = self.hidden, self.input_size, self.classes
        out = {}
        for d in ("l0_", "r0_"):
            out[self.rnn_prefix + d + "i2h_weight"] = self.get(self.rnn_prefix + d + "i2h_weight").reshape(self.gates * h, f)
            out[self.rnn_prefix + d + "h2h_weight"] = self.get(self.rnn_prefix + d + "h2h_weight").reshape(self.gates * h, h)
            out[self.rnn_prefix + d + "i2h_bias"] = self.get(self.rnn_prefix + d + "i2h_bias")
            out[self.rnn_prefix + d + "h2h_bias"] = self.get(self.rnn_prefix + d + "h2h_bias")
        out[self.dense_prefix + "weight"] = self.get(self.dense_prefix + "weight").reshape(c, 2 * h)
        out[self.dense_prefix + "bias"] = self.get(self.dense_prefix + "bias")
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_head_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class GNMTTrainer:
    """One training step of the captioner the way reference train_gnmt.py::train drives it (:328-337): teacher-forced
    ``NMTModel`` forward, token-averaged ``MaskedSoftmaxCELoss``, ``loss.backward()``, ``gluon.Trainer('adam').step(1)``.
    GRU (the reference's flag default) or LSTM cells; ``num_layers`` / ``num_bi_layers`` / ``use_residual`` as the reference's
    flags pass them into the model it trains (train_gnmt.py:58-61,223-227; round 4: any ``num_layers >= 2`` with
    ``num_bi_layers < num_layers``).  ``grads`` / ``params`` are flat device views for a data-parallel all-reduce between
    ``forward_backward`` and ``step``."""

    def __init__(self, params: dict, input_size: int, hidden: int, embed: int, vocab: int, max_batch: int = 32,
                 max_src_len: int = 256, max_tgt_len: int = 64, prefix: str = "gnmt_", ctx: _lib.Context | None = None,
                 cell_type: str = "gru", num_layers: int = 2, num_bi_layers: int = 1, use_residual: bool = False):
        if cell_type not in ("gru", "lstm"):
            raise ValueError(f"cell_type must be 'gru' or 'lstm', got {cell_type!r}")
        self.num_layers, self.num_bi_layers, self.use_residual = num_layers, num_bi_layers, bool(use_residual)
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.input_size, self.hidden, self.embed, self.vocab, self.prefix = input_size, hidden, embed, vocab, prefix
        self.cell_type = cell_type
        self.names = [k for k in params if k.startswith(prefix)]
        self.shapes = {k: tuple(np.asarray(params[k]).shape) for k in self.names}
        arr, keep = _lib.make_params({k: params[k] for k in self.names})
        h = C.c_void_p()
        check(self.lib.tn_gnmt_trainer_create_ex(self.ctx.handle, arr, len(arr), prefix.encode(),
                                                 _lib.RNN_GRU if cell_type == "gru" else _lib.RNN_LSTM, input_size, hidden, embed, vocab,
                                                 num_layers, num_bi_layers, 1 if use_residual else 0,
                                                 max_batch, max_src_len, max_tgt_len, C.byref(h)), "tn_gnmt_trainer_create_ex")
        del keep
        self.handle = h
        pw, pg, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.tn_gnmt_trainer_buffers(h, C.byref(pw), C.byref(pg), C.byref(n)), "tn_gnmt_trainer_buffers")
        self.numel, self._pw, self._pg = n.value, pw.value, pg.value

    def _view(self, addr):
        class _Arr:
            __cuda_array_interface__ = {"shape": (self.numel,), "typestr": "<f4", "data": (addr, False), "version": 3}
        return torch.as_tensor(_Arr(), device=f"cuda:{self.ctx.device}")

    @property
    def grads(self) -> torch.Tensor:
        return self._view(self._pg)

    @property
    def params(self) -> torch.Tensor:
        return self._view(self._pw)

    def forward_backward(self, src: torch.Tensor, src_valid_length: torch.Tensor, tgt: torch.Tensor,
                         tgt_valid_length: torch.Tensor, return_logits: bool = False):
        """src (B,T,F) fp32, tgt (B,L) token ids incl. BOS / EOS, valid lengths (B,) -> loss (0-d tensor) [, logits (B,L-1,V)]"""
        src = src.contiguous().float()
        b, t, _ = src.shape
        dev = src.device
        tgt = tgt.to(device=dev, dtype=torch.int32).contiguous()
        svl = src_valid_length.to(device=dev, dtype=torch.int32).contiguous()
        tvl = tgt_valid_length.to(device=dev, dtype=torch.int32).contiguous()
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        logits = torch.empty((b, tgt.shape[1] - 1, self.vocab), dtype=torch.float32, device=dev) if return_logits else None
        check(self.lib.tn_gnmt_trainer_forward_backward(self.handle, ptr(src), ptr(svl), ptr(tgt), tgt.shape[1], ptr(tvl), b, t,
                                                        tgt.shape[1], ptr(loss), ptr(logits)), "tn_gnmt_trainer_forward_backward")
        return (loss[0], logits) if return_logits else loss[0]

    def set_dropout(self, p: float, seed: int = 0):
        """``--dropout`` of train_gnmt.py (default 0.2 there): after each encoder layer and on the top decoder cell's output."""
        check(self.lib.tn_gnmt_trainer_set_dropout(self.handle, p, seed), "tn_gnmt_trainer_set_dropout")

    def dropout_masks(self, batch: int, src_steps: int, tgt_steps: int):
        """The last step's masks as tensors: (B,T,2H), (B,T,H), (L,B,H) (step-major) - for tests against the oracle."""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.lib.tn_gnmt_trainer_dropout_masks(self.handle, C.byref(a), C.byref(b), C.byref(c)), "tn_gnmt_trainer_dropout_masks")
        h = self.hidden

        def view(addr, shape):
            class _Arr:
                __cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (addr, False), "version": 3}
            return torch.as_tensor(_Arr(), device=f"cuda:{self.ctx.device}")
        return (view(a.value, (batch, src_steps, 2 * h)), view(b.value, (batch, src_steps, h)), view(c.value, (tgt_steps, batch, h)))

    def dropout_mask(self, which: int, shape) -> torch.Tensor:
        """One mask of the last step: ``which`` = encoder layer ``i`` -> (B,T,dirs*H); ``num_layers + j`` -> decoder layer ``j >= 1``,
        (L,B,H) step-major."""
        a = C.c_void_p()
        check(self.lib.tn_gnmt_trainer_dropout_mask(self.handle, which, C.byref(a)), "tn_gnmt_trainer_dropout_mask")

        class _Arr:
            __cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (a.value, False), "version": 3}
        return torch.as_tensor(_Arr(), device=f"cuda:{self.ctx.device}")

    def step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, epsilon: float = 1e-8):
        check(self.lib.tn_gnmt_trainer_adam_step(self.handle, lr, beta1, beta2, epsilon), "tn_gnmt_trainer_adam_step")

    def get(self, name: str, gradient: bool = False) -> np.ndarray:
        shape = self.shapes[name]
        out = np.empty(int(np.prod(shape)), np.float32)
        n = C.c_int64()
        check(self.lib.tn_gnmt_trainer_read_param(self.handle, name.encode(), 1 if gradient else 0,
                                                  out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(n)),
              "tn_gnmt_trainer_read_param")
        return out[:n.value].reshape(shape).copy()

    def state_dict(self) -> dict:
        return {k: self.get(k) for k in self.names}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_gnmt_trainer_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class FrameModelTrainer:
    """End-to-end fine-tuning step of ``FrameModel(DenseNet121.features, classes)`` the way reference train.py drives it
    with an un-frozen backbone: BatchNorm in training mode, ``SoftmaxCrossEntropyLoss`` per sample (:324), backward of the
    summed losses (:419-421), ``gluon.Trainer('sgd', {lr, momentum, wd}).step(batch_size)`` (:298-299,424).  fp32.  The batch
    size is fixed at construction (BatchNorm statistics are per batch)."""

    def __init__(self, params: dict, size: int = 224, classes: int = 11, batch: int = 8, prefix: str = "densenet0_",
                 dense_prefix: str = "framemodel0_dense0_", ctx: _lib.Context | None = None):
        self.ctx = ctx or _lib.default_context()
        self.lib = self.ctx.lib
        self.size, self.classes, self.batch = size, classes, batch
        self.names = [k for k in params if k.startswith(prefix) or k.startswith(dense_prefix)]
        self.shapes = {k: tuple(np.asarray(params[k]).shape) for k in self.names}
        arr, keep = _lib.make_params({k: params[k] for k in self.names})
        h = C.c_void_p()
        check(self.lib.tn_finetune_create(self.ctx.handle, arr, len(arr), prefix.encode(), dense_prefix.encode(), size, size, classes,
                                          batch, C.byref(h)), "tn_finetune_create")
        del keep
        self.handle = h
        pw, pg, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(self.lib.tn_finetune_buffers(h, C.byref(pw), C.byref(pg), C.byref(n)), "tn_finetune_buffers")
        self.numel, self._pw, self._pg = n.value, pw.value, pg.value

    def _view(self, addr):
        class _Arr:
            __cuda_array_interface__ = {"shape": (self.numel,), "typestr": "<f4", "data": (addr, False), "version": 3}
        return torch.as_tensor(_Arr(), device=f"cuda:{self.ctx.device}")

    @property
    def grads(self) -> torch.Tensor:
        return self._view(self._pg)

    def forward_backward(self, x: torch.Tensor, labels: torch.Tensor):
        """x: frames as NCHW fp32 (the reference layout) or NHWC fp32, normalised; labels (B,) -> (loss (B,), logits (B, classes))"""
        _on_ctx_device(self.ctx, x, "FrameModelTrainer")
        sz = self.size
        if x.dtype == torch.uint8:        # decoded frames out of transforms.Compose: ToTensor + Normalize here
            x = to_tensor_normalize(x, ctx=self.ctx)
        if x.dim() == 4 and tuple(x.shape[1:]) == (3, sz, sz):
            x = x.permute(0, 2, 3, 1)
        if x.dim() != 4 or tuple(x.shape) != (self.batch, sz, sz, 3):
            raise ValueError(f"FrameModelTrainer expects ({self.batch}, 3, {sz}, {sz}) or ({self.batch}, {sz}, {sz}, 3) frames "
                             f"(apply the Resize/CenterCrop transform first), got {tuple(x.shape)}")
        x = x.contiguous().float()
        b = x.shape[0]
        labels = labels.to(device=x.device, dtype=torch.int32).contiguous()
        loss = torch.empty((b,), dtype=torch.float32, device=x.device)
        logits = torch.empty((b, self.classes), dtype=torch.float32, device=x.device)
        check(self.lib.tn_finetune_forward_backward(self.handle, ptr(x), ptr(labels), b, sz, sz, ptr(loss), ptr(logits)),
              "tn_finetune_forward_backward")
        return loss, logits

    def step(self, batch_size: int, lr: float, momentum: float = 0.9, wd: float = 1e-4):
        check(self.lib.tn_finetune_sgd_step(self.handle, lr, momentum, wd, 1.0 / batch_size), "tn_finetune_sgd_step")

    def get(self, name: str, gradient: bool = False, shape=None) -> np.ndarray:
        shape = shape or self.shapes[name]
        out = np.empty(int(np.prod(shape)), np.float32)
        n = C.c_int64()
        check(self.lib.tn_finetune_read_param(self.handle, name.encode(), 1 if gradient else 0,
                                              out.ctypes.data_as(C.POINTER(C.c_float)), out.size, C.byref(n)), "tn_finetune_read_param")
        return out[:n.value].reshape(shape).copy()

    def state_dict(self) -> dict:
        return {k: self.get(k) for k in self.names}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.tn_finetune_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
