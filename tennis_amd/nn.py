"""Gluon-named leaf blocks backed by the HIP library: Dense, GRU/LSTM layers and
the DenseNet-121 ``.features`` backbone (model_zoo.get_model)."""
from __future__ import annotations

import numpy as np
import torch

from . import engine, weights as W
from .block import Block, Parameter


def _to_device(x):
    """Accept numpy or torch (cpu/cuda); return a cuda tensor (device plumbing only)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x if x.is_cuda else x.cuda()


class Dense(Block):
    """``nn.Dense(units, flatten=True)`` with deferred input size (reference definitions.py:25)."""

    def __init__(self, units, flatten=True, in_units=0, seed=1, **kwargs):
        super().__init__(hint="dense", **kwargs)
        self._units, self._in_units, self._seed = units, in_units, seed
        self._own_params[self.prefix + "weight"] = Parameter(self.prefix + "weight")
        self._own_params[self.prefix + "bias"] = Parameter(self.prefix + "bias")

    def _materialize(self, in_units):
        w = self._own_params[self.prefix + "weight"]
        if w.data is None:
            p = W.make_dense_weights(self._seed, self._units, in_units, self.prefix)
            self._adopt(p)
        self._in_units = w.data.shape[1]

    def forward(self, x):
        x = _to_device(x)
        x = x.reshape(x.shape[0], -1)
        self._materialize(x.shape[1])
        if self._engine is None:
            self._engine = engine.Dense(self._own_params[self.prefix + "weight"].data,
                                        self._own_params[self.prefix + "bias"].data)
        return self._engine(x)


class _RNNLayer(Block):
    _mode = "gru"

    def __init__(self, hidden_size, num_layers=1, layout="NTC", bidirectional=False, input_size=0, seed=2,
                 **kwargs):
        super().__init__(hint=self._mode, **kwargs)
        if num_layers != 1 or layout != "NTC":
            raise NotImplementedError("only single-layer NTC recurrent layers are on the hot path "
                                      "(reference definitions.py:94-96)")
        self._hidden, self._bi, self._input_size, self._seed = hidden_size, bidirectional, input_size, seed
        for d in (["l", "r"] if bidirectional else ["l"]):
            for n in ("i2h_weight", "h2h_weight", "i2h_bias", "h2h_bias"):
                k = f"{self.prefix}{d}0_{n}"
                self._own_params[k] = Parameter(k)

    def _materialize(self, f):
        if next(iter(self._own_params.values())).data is None:
            self._adopt(W.make_rnn_weights(self._seed, self._mode, f, self._hidden, self.prefix, self._bi))
        self._input_size = self._own_params[f"{self.prefix}l0_i2h_weight"].data.shape[1]

    def forward(self, x, valid_length=None):
        x = _to_device(x)
        b, t, f = x.shape
        self._materialize(f)
        if self._engine is None or self._engine.max_rows < b * t:
            p = {k: v.data for k, v in self._own_params.items()}
            self._engine = engine.BiRNN(self._mode, self._input_size, self._hidden, p, self.prefix, self._bi,
                                        max_rows=max(b * t, 4096))
        return self._engine(x, valid_length)


class GRU(_RNNLayer):
    """``mx.gluon.rnn.GRU`` (reference definitions.py:96)."""
    _mode = "gru"


class LSTM(_RNNLayer):
    """``mx.gluon.rnn.LSTM`` (reference definitions.py:94)."""
    _mode = "lstm"


class DenseNet121Backbone(Block):
    """``get_model('DenseNet121', ...).features`` (reference evaluate.py:125): frames -> (B, F) fp32."""

    def __init__(self, seed=0, prefix="densenet0_", max_batch=256, exact_weights=False, conversion="nearest", **kwargs):
        """How adopted fp32 conv weights become the fp16 model the kernels evaluate:
        ``conversion="nearest"`` (default): rounded once to the nearest fp16 on adoption;
        ``conversion="calibrated"``: kept in fp32 until ``calibrate()`` - or the first forward - and then rounded with vector error
        feedback against the mean activations of the built-in calibration frames (tennis_amd.calibrate; the same frames, hence
        the same model, on every rank of a multi-GPU job; ``calibrate(frames)`` adds uint8 frames of the footage): one fp16 number
        per weight, full speed, features within 1e-3 of the fp32 evaluation on natural content (per-family figures: DESIGN.md);
        ``exact_weights=True`` / ``conversion="exact"``: conv weights stay fp32 and the library evaluates them as hi + lo
        fp16 pairs (engine.DenseNet121Features(exact_weights=True)) at twice the MFMAs."""
        super().__init__(prefix=prefix, **kwargs)
        if conversion not in ("nearest", "calibrated", "exact"):
            raise ValueError(f"conversion must be 'nearest', 'calibrated' or 'exact', got {conversion!r}")
        self._seed, self._max_batch, self._exact = seed, max_batch, bool(exact_weights) or conversion == "exact"
        self._calibrated_mode, self._converted = conversion == "calibrated" and not self._exact, None
        convs, final_bn, cfin = W.densenet121_layout()
        names = []
        for cv in convs:
            names.append(prefix + cv["name"] + "_weight")
            names += [prefix + cv["bn"] + s for s in ("_gamma", "_beta", "_running_mean", "_running_var")]
        names += [prefix + final_bn + s for s in ("_gamma", "_beta", "_running_mean", "_running_var")]
        for n in names:
            self._own_params[n] = Parameter(n)
        self._size = None

    def initialize(self, *a, **k):
        super().initialize(*a, **k)
        if next(iter(self._own_params.values())).data is None:
            self._adopt(W.make_densenet121_weights(self._seed, self.prefix, fp16_model=not (self._exact or self._calibrated_mode)))

    def _structural_params(self, path: str = "") -> dict:
        """Structural names of gluon ``model_zoo.vision.densenet121().features`` [EXT]: a HybridSequential of
        0 conv 7x7, 1 BatchNorm, 2 relu, 3 maxpool, then (dense block, transition) pairs at 4..10, 11 BatchNorm,
        12 relu, 13 avgpool, 14 flatten.  A dense block is a HybridSequential of layers; a layer is
        HybridConcurrent[0 Identity, 1 HybridSequential(0 BN, 1 relu, 2 conv 1x1, 3 BN, 4 relu, 5 conv 3x3)];
        a transition is HybridSequential(0 BN, 1 relu, 2 conv 1x1, 3 avgpool)."""
        bn = ("gamma", "beta", "running_mean", "running_var")
        out = {}
        pre = self.prefix

        def put_bn(spath, pname):
            for s_ in bn:
                out[f"{path}{spath}.{s_}"] = f"{pre}{pname}_{s_}"
        out[f"{path}0.weight"] = pre + "conv0_weight"
        put_bn("1", "batchnorm0")
        child, outer = 4, 1
        for b, nl in enumerate((6, 12, 24, 16)):
            sp = f"stage{b + 1}_"
            for l in range(nl):
                put_bn(f"{child}.{l}.1.0", f"{sp}batchnorm{2 * l}")
                out[f"{path}{child}.{l}.1.2.weight"] = f"{pre}{sp}conv{2 * l}_weight"
                put_bn(f"{child}.{l}.1.3", f"{sp}batchnorm{2 * l + 1}")
                out[f"{path}{child}.{l}.1.5.weight"] = f"{pre}{sp}conv{2 * l + 1}_weight"
            child += 1
            if b < 3:
                put_bn(f"{child}.0", f"batchnorm{outer}")
                out[f"{path}{child}.2.weight"] = f"{pre}conv{outer}_weight"
                child += 1
                outer += 1
        put_bn(str(child), f"batchnorm{outer}")
        assert set(out.values()) == set(self._own_params), "structural table out of sync with the layout"
        return out

    def _adopt(self, params):
        own = {k: v for k, v in params.items() if k in self._own_params}
        keep_fp32 = self._exact or self._calibrated_mode
        super()._adopt(own if keep_fp32 else W.as_fp16_model(own))
        self._converted = None

    def calibrate(self, frames=None, size=224):
        """``conversion="calibrated"``: measure, frame by frame, the mean activation of every convolution input on the built-in
        calibration set (+ ``frames``: NHWC uint8 frames of the material to be processed, if given) and convert the fp32 weights
        against them.  Deterministic and rank-independent: every rank of a sharded job that calls this with the same ``frames``
        (or none) serves the same fp16 model."""
        from .calibrate import calibrated_fp16_model
        if frames is not None:
            frames = _to_device(frames)
            if frames.dtype != torch.uint8:
                raise TypeError("calibrate(frames): NHWC uint8 frames (as the loader hands them over)")
            size = tuple(frames.shape[1:3])
        self._engine = None
        p = {k: v.data for k, v in self._own_params.items()}
        self._converted = calibrated_fp16_model(p, frames, size, prefix=self.prefix)

    def forward(self, x):
        x = _to_device(x)
        if next(iter(self._own_params.values())).data is None:
            self.initialize()
        if x.dim() != 4:
            raise ValueError(f"backbone expects a 4-d frame batch, got shape {tuple(x.shape)}")
        size = tuple(x.shape[2:]) if x.shape[1] == 3 and x.dtype == torch.float32 else tuple(x.shape[1:3])
        b = x.shape[0]
        if self._engine is None or self._size != size or self._engine.max_batch < b:
            self._engine = None  # release the old workspace first
            if self._calibrated_mode and self._converted is None:
                self.calibrate(None, size)      # the built-in set only: never this rank's own first frames
            p = self._converted if self._calibrated_mode else {k: v.data for k, v in self._own_params.items()}
            self._engine = engine.DenseNet121Features(p, size, max_batch=max(b, min(self._max_batch, 64)),
                                                      prefix=self.prefix, exact_weights=self._exact)
            self._size = size
        return self._engine(x)
