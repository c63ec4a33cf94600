"""MXNet ``.params`` container reader / writer (SURVEY §8f-2, Appendix B "MXNet .params").

The reference saves and loads weights with Gluon's ``save_parameters`` / ``load_parameters``
(evaluate.py:198,212,239; train.py:497), i.e. ``mx.nd.save`` of a name -> NDArray dict.  Layout as published in
MXNet 1.x (src/ndarray/ndarray.cc NDArray::Save/Load, src/c_api/c_api.cc MXNDArraySave) - PARITY UNPINNED: no
``.params`` file ships with the reference, so this follows the format description only and is tested by round trip:

    uint64 0x112 (list magic) | uint64 reserved | uint64 n_arrays
    n_arrays x { uint32 magic (0xF993FAC9 V2 | 0xF993FACA V3 | 0xF993FAC8 V1 | none = legacy)
                 [V2/V3] int32 storage type (0 = dense; sparse is refused)
                 shape: uint32 ndim, ndim x int64 (legacy / V1: ndim x uint32)
                 int32 dev_type, int32 dev_id | int32 dtype flag | raw little-endian data }
    uint64 n_names | n_names x { uint64 length, bytes }

Names are returned as stored except for the ``arg:`` / ``aux:`` prefixes of Module checkpoints.  Gluon's
*structural* names (``features.0.weight``) need the model tree of the saving package to be mapped onto the
prefixed names this package uses (``densenet0_conv0_weight``); files written with
``collect_params().save`` / ``mx.nd.save`` of prefixed names load directly.
"""
from __future__ import annotations

import struct

import numpy as np

LIST_MAGIC = 0x112
V1, V2, V3 = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
DTYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
FLAGS = {np.dtype(v): k for k, v in DTYPES.items()}


def is_mxnet_params(path) -> bool:
    with open(path, "rb") as f:
        head = f.read(8)
    return len(head) == 8 and struct.unpack("<Q", head)[0] == LIST_MAGIC


def _read_array(f):
    (magic,) = struct.unpack("<I", f.read(4))
    if magic in (V2, V3):
        (stype,) = struct.unpack("<i", f.read(4))
        if stype != 0:
            raise ValueError("sparse NDArray storage is not supported")
        (ndim,) = struct.unpack("<I" if magic == V2 else "<i", f.read(4))
        if magic == V3 and ndim < 0:
            return None
        shape = struct.unpack(f"<{ndim}q", f.read(8 * ndim)) if ndim else ()
    elif magic == V1:
        (ndim,) = struct.unpack("<I", f.read(4))
        shape = struct.unpack(f"<{ndim}q", f.read(8 * ndim)) if ndim else ()
    else:                                   # legacy: the word just read is ndim, dims are uint32
        ndim = magic
        shape = struct.unpack(f"<{ndim}I", f.read(4 * ndim)) if ndim else ()
    if ndim == 0 and magic != V3:
        return None                         # "none" array: nothing else stored
    f.read(8)                               # context (dev_type, dev_id): always loaded to host
    (flag,) = struct.unpack("<i", f.read(4))
    if flag not in DTYPES:
        raise ValueError(f"unknown dtype flag {flag}")
    dt = np.dtype(DTYPES[flag]).newbyteorder("<")
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return data.reshape(shape).astype(DTYPES[flag])


def load_mxnet_params(path) -> dict:
    """-> {name: ndarray}; unnamed lists come back as {'0': ..., '1': ...}."""
    with open(path, "rb") as f:
        magic, _ = struct.unpack("<QQ", f.read(16))
        if magic != LIST_MAGIC:
            raise ValueError(f"{path}: not an MXNet NDArray list (magic {magic:#x})")
        (n,) = struct.unpack("<Q", f.read(8))
        arrays = [_read_array(f) for _ in range(n)]
        (nn,) = struct.unpack("<Q", f.read(8))
        names = []
        for _ in range(nn):
            (ln,) = struct.unpack("<Q", f.read(8))
            names.append(f.read(ln).decode("utf-8"))
    if nn and nn != n:
        raise ValueError(f"{path}: {n} arrays but {nn} names")
    if not nn:
        names = [str(i) for i in range(n)]
    out = {}
    for k, a in zip(names, arrays):
        if a is None:
            continue
        for pre in ("arg:", "aux:"):
            if k.startswith(pre):
                k = k[len(pre):]
        out[k] = a
    return out


def save_mxnet_params(path, params: dict) -> None:
    """Write {name: ndarray} as an NDArray list (V2 records, dense, cpu(0))."""
    with open(path, "wb") as f:
        f.write(struct.pack("<QQQ", LIST_MAGIC, 0, len(params)))
        for a in params.values():
            a = np.ascontiguousarray(a)
            if a.dtype not in FLAGS:
                raise ValueError(f"dtype {a.dtype} has no MXNet type flag")
            f.write(struct.pack("<IiI", V2, 0, a.ndim))
            f.write(struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(struct.pack("<iii", 1, 0, FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder("<")).tobytes())
        f.write(struct.pack("<Q", len(params)))
        for k in params:
            b = k.encode("utf-8")
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
