"""Minimal stand-in for the Gluon ``HybridBlock`` surface the reference scripts use.

The reference drives its models through a handful of Gluon methods
(SURVEY §8b): ``initialize()`` (evaluate.py:166), ``summary(x)`` (:173-181),
``collect_params().reset_ctx(ctx)`` / ``.values()[i].grad_req`` (:183,153-154),
``hybridize()`` (:184), ``load_parameters(path[, ctx])`` (:198,212,239),
``save_parameters`` (train.py:497) and attribute access to children.  This module
keeps those names and meanings; the compute behind ``__call__`` is the HIP
library.  Parameters are fp32 numpy arrays keyed by Gluon names and are stored
in MXNet ``.params`` containers under Gluon's structural names (what ``save_parameters``
of the reference writes) or ``.npz`` under the prefixed names; both load back.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np


class Parameter:
    def __init__(self, name, data=None):
        self.name = name
        self.data = data
        self.grad_req = "write"

    @property
    def shape(self):
        return None if self.data is None else self.data.shape


class ParameterDict(OrderedDict):
    """``collect_params()`` result: name -> Parameter."""

    def reset_ctx(self, ctx):  # device placement happens when the engine handle is built
        self.ctx = ctx

    def setattr(self, name, value):
        for p in self.values():
            setattr(p, name, value)


class Block:
    _counters: dict = {}

    def __init__(self, prefix: str | None = None, hint: str | None = None):
        hint = hint or type(self).__name__.lower()
        if prefix is None:
            n = Block._counters.get(hint, 0)
            Block._counters[hint] = n + 1
            prefix = f"{hint}{n}_"
        object.__setattr__(self, "_children", OrderedDict())
        self.prefix = prefix
        self._own_params = OrderedDict()
        self._engine = None
        self._initialized = False

    # -- tree ------------------------------------------------------------
    def __setattr__(self, key, value):
        if isinstance(value, Block) and key != "_parent":
            self._children[key] = value
        object.__setattr__(self, key, value)

    def name_scope(self):
        return _NullScope()

    def _all_blocks(self):
        seen, out = set(), []

        def walk(b):
            if id(b) in seen:
                return
            seen.add(id(b))
            out.append(b)
            for c in b._children.values():
                walk(c)
        walk(self)
        return out

    # -- parameters --------------------------------------------------------
    def collect_params(self) -> ParameterDict:
        pd = ParameterDict()
        for b in self._all_blocks():
            for k, v in b._own_params.items():
                pd[k] = v
        return pd

    def initialize(self, init=None, ctx=None, verbose=False, force_reinit=False):
        for b in self._all_blocks():
            b._initialized = True

    def hybridize(self, active=True, **kwargs):
        return None

    def _invalidate(self):
        for b in self._all_blocks():
            b._engine = None

    def set_params(self, params: dict):
        """Adopt arrays for every known/deferred parameter name under this tree."""
        for b in self._all_blocks():
            b._adopt(params)
        self._invalidate()

    def _adopt(self, params):
        for k in list(self._own_params):
            if k in params:
                self._own_params[k].data = np.ascontiguousarray(params[k], dtype=np.float32)

    # -- structural names ----------------------------------------------------
    def _structural_params(self, path: str = "") -> dict:
        """``Block._collect_params_with_prefix`` of Gluon [EXT]: the names ``save_parameters`` / ``load_parameters``
        use — attribute path of the child blocks (``backbone``, ``classes``, ``td.model``, ``rnn``; sequential
        children by index), then the parameter's own name without the block prefix (``weight``, ``gamma``,
        ``l0_i2h_weight`` ...).  -> {structural name: prefixed name}."""
        out = {}
        for k in self._own_params:
            out[path + (k[len(self.prefix):] if k.startswith(self.prefix) else k)] = k
        for name, child in self._children.items():
            out.update(child._structural_params(path + name + "."))
        return out

    def _structural_transposed(self, path: str = "") -> set:
        """Structural names whose array is stored transposed in a Gluon checkpoint relative to the engine's own
        parameter (only the captioner's attention projection, models/captioning/gnmt.py)."""
        out = set()
        for name, child in self._children.items():
            out |= child._structural_transposed(path + name + ".")
        return out

    def save_parameters(self, filename, structural=None):
        """``<name>.params`` (the only form the reference writes, train.py:497) -> the MXNet NDArray-list container
        with Gluon's structural names, which ``mx.gluon.Block.load_parameters`` of the reference reads back;
        any other extension -> ``.npz`` with the prefixed names."""
        pd = {k: v.data for k, v in self.collect_params().items() if v.data is not None}
        if structural is None:
            structural = str(filename).endswith(".params")
        if structural:
            from .params_io import save_mxnet_params
            names = {}
            for k, v in self._structural_params().items():
                names.setdefault(v, k)          # the first structural name of a parameter is the one Gluon writes
            tr = self._structural_transposed()
            save_mxnet_params(filename, {names[k]: (np.ascontiguousarray(a.T) if names[k] in tr else a)
                                         for k, a in pd.items()})
            return
        with open(filename, "wb") as f:
            np.savez(f, **pd)

    def load_parameters(self, filename, ctx=None, allow_missing=False, ignore_extra=False):
        if not os.path.exists(filename):
            raise FileNotFoundError(filename)
        from .params_io import is_mxnet_params, load_mxnet_params
        if is_mxnet_params(filename):       # an MXNet NDArray list (mx.nd.save / Gluon save with prefixed names)
            loaded = load_mxnet_params(filename)
        else:
            with np.load(filename) as z:
                loaded = {k: z[k] for k in z.files}
        smap = self._structural_params()
        known = set(self.collect_params())
        if any(k in smap and k not in known for k in loaded):       # Gluon save_parameters: structural names
            tr = self._structural_transposed()
            loaded = {smap.get(k, k): (np.ascontiguousarray(v.T) if k in tr else v) for k, v in loaded.items()}
        extra = [k for k in loaded if k not in known]
        if extra and not ignore_extra:
            raise AssertionError(f"Parameter '{extra[0]}' loaded from file '{filename}' is not present in this block")
        self.set_params(loaded)
        if not allow_missing:
            missing = [k for k, v in self.collect_params().items() if v.data is None and k not in loaded]
            if missing:
                raise AssertionError(f"Parameter '{missing[0]}' is missing in file '{filename}'")

    def summary(self, *inputs):
        lines = [f"{type(self).__name__} ({self.prefix})"]
        for k, v in self.collect_params().items():
            lines.append(f"  {k:<48} {None if v.data is None else tuple(v.data.shape)}")
        n = sum(v.data.size for v in self.collect_params().values() if v.data is not None)
        lines.append(f"  total parameters: {n}")
        return "\n".join(lines)

    # -- call --------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, *args, **kwargs):  # pragma: no cover
        raise NotImplementedError


class _NullScope:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
