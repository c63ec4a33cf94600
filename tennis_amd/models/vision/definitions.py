"""Model specifications — mirror of reference models/vision/definitions.py.

Same class names, constructor arguments and attributes (``backbone``, ``classes``,
``td``, ``rnn``) as the reference; ``model(x)`` runs on the MI355X through
libtennis_hip.so.  Inputs may be numpy or torch tensors; outputs are torch CUDA
tensors (fp32), the counterpart of MXNet NDArrays on ``mx.gpu``.
"""
from ...block import Block
from ...engine import temporal_pool
from ...nn import GRU, LSTM, Dense, _to_device
from ...utils.layers import TimeDistributed


class FrameModel(Block):
    """Reference definitions.py:10-33: backbone CNN + one Dense to the classes."""

    def __init__(self, backbone, num_classes=-1, swap=False, **kwargs):
        super().__init__(**kwargs)
        if swap:
            raise NotImplementedError("swap=True is the R(2+1)D path (reference evaluate.py:132), out of scope")
        self.swap = swap
        self.backbone = backbone
        self.classes = None
        if num_classes > 0:
            self.classes = Dense(num_classes, flatten=True, prefix=self.prefix + "dense0_")

    def forward(self, x):
        x = self.backbone(x)            # definitions.py:30
        if self.classes:
            x = self.classes(x)         # definitions.py:31-32
        return x


class TemporalPooling(Block):
    """Reference definitions.py:36-72."""

    def __init__(self, model, num_classes=-1, pool="max", feats=False, **kwargs):
        super().__init__(**kwargs)
        self.pool = pool
        self.feats = feats
        self.classes = None
        if model is not None:
            if num_classes == 0:                       # definitions.py:53-55
                self.td = TimeDistributed(model.backbone)
                self.classes = model.classes
            else:                                      # definitions.py:56-59
                self.td = TimeDistributed(model)
                if num_classes > 0:
                    self.classes = Dense(num_classes, flatten=True, prefix=self.prefix + "dense0_")
        else:                                          # definitions.py:60-61
            self.classes = Dense(num_classes, flatten=True, prefix=self.prefix + "dense0_")

    def forward(self, x):
        if not self.feats:
            x = self.td(x)                             # definitions.py:64-65
        x = temporal_pool(_to_device(x), "mean" if self.pool == "mean" else "max")   # :66-69
        if self.classes:
            x = self.classes(x)
        return x


class CNNRNN(Block):
    """Reference definitions.py:75-110: [TimeDistributed CNN ->] bi-GRU/LSTM -> max over T -> Dense."""

    def __init__(self, model, num_classes=-1, hidden_size=128, type="gru", **kwargs):
        super().__init__(**kwargs)
        self.feats = model is None
        if model is not None:
            self.td = TimeDistributed(model.backbone)                  # definitions.py:91-92
        if type == "lstm":                                             # definitions.py:93-96
            self.rnn = LSTM(hidden_size, layout="NTC", bidirectional=True, prefix=self.prefix + "lstm0_")
        else:
            self.rnn = GRU(hidden_size, layout="NTC", bidirectional=True, prefix=self.prefix + "gru0_")
        self.classes = None
        if num_classes == 0:                                           # definitions.py:98-101
            self.classes = model.classes
        elif num_classes > 0:
            self.classes = Dense(num_classes, flatten=True, prefix=self.prefix + "dense0_")

    def forward(self, x):
        if not self.feats:
            x = self.td(x)                                             # definitions.py:104-105
        x = self.rnn(x)                                                # :106
        x = temporal_pool(x, "max")                                    # :107
        if self.classes:
            x = self.classes(x)                                        # :108-109
        return x


class TwoStreamModel(Block):
    """Reference definitions.py:127-153 — optical-flow two-stream input is out of the
    hot path (needs 217 GB of flow JPEGs + FlowNet weights)."""

    def __init__(self, *a, **k):
        raise NotImplementedError("TwoStreamModel (flow input) is outside the accelerated hot path")
