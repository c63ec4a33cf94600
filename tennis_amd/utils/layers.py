"""Custom layers — mirror of reference utils/layers.py."""
from ..block import Block


class TimeDistributed(Block):
    """Reference utils/layers.py:9-48.  Applies ``model`` to every timestep by
    folding time into the batch axis ('reshape' style, layers.py:38-46): the
    device sees one (B*T, ...) batch, so this is a view change, not a kernel
    (SURVEY §2c K8).  The 'for' style (layers.py:27-36) gives the same result
    and is executed the same way here."""

    def __init__(self, model, style="reshape", **kwargs):
        super().__init__(**kwargs)
        assert style in ["reshape", "for"]
        self._style = style
        self.model = model

    def forward(self, x):
        b, t = x.shape[0], x.shape[1]
        y = self.model(x.reshape((b * t,) + tuple(x.shape[2:])))   # layers.py:39-40
        if isinstance(y, tuple):                                   # layers.py:41-44
            return tuple(yi.reshape((b, t) + tuple(yi.shape[1:])) for yi in y)
        if isinstance(y, list):
            return [yi.reshape((b, t) + tuple(yi.shape[1:])) for yi in y]
        return y.reshape((b, t) + tuple(y.shape[1:]))              # layers.py:46
