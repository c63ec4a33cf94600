"""``mx.image`` surface of the reference's frame loader, on the GPU.

The reference reads every frame with ``mx.image.imread(path, 1)`` (dataset.py:204,216): OpenCV's imdecode on a DataLoader
worker, i.e. libjpeg on the host, one process per CPU core (train.py:101-102, ``num_workers = cpu_count``) - the real-data
bottleneck of the pipeline.  Here the files' bytes go to the device as they are and ``libtennis_hip``'s ``tn_jpeg_decode``
decodes a whole batch there (parallel Huffman decoding, libjpeg's integer IDCT / fancy upsampling / colour conversion,
bit-exact with Pillow = libjpeg-turbo; csrc/jpeg.hip):

    imdecode_batch(list_of_bytes)      -> (N, H, W, 3) uint8 RGB torch tensor on the GPU
    imread_batch(list_of_paths)        -> same, reading the files
    imdecode(buf, flag=1) / imread(path, flag=1)   one image, (H, W, 3)
    image_info(buf)                    -> (width, height, components)   header only, host

One call decodes files of one geometry (the frames of a video).  What the device decoder does not handle (progressive,
arithmetic, 12-bit, CMYK, multi-scan files, corrupt data) raises ``UnsupportedJpeg`` (a ``RuntimeError``) naming the file -
there is no silent CPU fallback; a caller that wants one (``TennisSet(decode="auto")``) asks for it explicitly, and it only
covers that refusal: a missing library or GPU still raises.
"""
from __future__ import annotations

import atexit
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

__all__ = ["JpegDecoder", "UnsupportedJpeg", "imdecode", "imdecode_batch", "imread", "imread_batch", "image_info"]


class UnsupportedJpeg(RuntimeError):
    """the device decoder refused the input (TN_ERR_INVALID: a JPEG process it does not decode, mixed geometry, corrupt data)"""


def image_info(buf) -> tuple:
    """(width, height, components) of a JPEG byte string, from its header (host only)."""
    buf = bytes(buf)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    lib = _lib.load()
    rc = lib.tn_jpeg_info(buf, len(buf), C.byref(w), C.byref(h), C.byref(c), None, None)
    if rc == -1:
        raise UnsupportedJpeg("libtennis_hip tn_jpeg_info refused the input: " + lib.tn_last_error().decode("utf-8", "replace"))
    check(rc, "tn_jpeg_info")
    return w.value, h.value, c.value


_live = weakref.WeakSet()          # decoders that still own device / pinned memory


@atexit.register
def _close_all():
    # release the workspaces while the HIP runtime is still up (an object that survives until interpreter teardown -
    # e.g. one kept alive by a traceback - must not call into a runtime that has already shut down)
    for d in list(_live):
        d.close()


class JpegDecoder:
    """Owns a ``tn_jpeg`` workspace on one context (device + stream)."""

    def __init__(self, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.default_context()
        hd = C.c_void_p()
        check(self.ctx.lib.tn_jpeg_create(self.ctx.handle, C.byref(hd)), "tn_jpeg_create")
        self._h = hd
        _live.add(self)

    def decode(self, bufs, out: torch.Tensor | None = None) -> torch.Tensor:
        """bufs: sequence of bytes-like JPEG files of one geometry -> (N, H, W, 3) uint8 RGB on the context's device."""
        if self._h is None:
            raise RuntimeError("this JpegDecoder has been closed")
        bufs = [b if isinstance(b, bytes) else bytes(b) for b in bufs]
        n = len(bufs)
        if n == 0:
            raise ValueError("no files to decode")
        w, h, _ = image_info(bufs[0])
        dev = torch.device("cuda", self.ctx.device)
        if out is None:
            out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
        elif tuple(out.shape) != (n, h, w, 3) or out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous CUDA uint8 tensor of shape {(n, h, w, 3)}")
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in bufs])
        sizes = (C.c_size_t * n)(*[len(b) for b in bufs])
        ww, hh = C.c_int(), C.c_int()
        rc = self.ctx.lib.tn_jpeg_decode(self._h, ptrs, sizes, n, ptr(out), C.byref(ww), C.byref(hh))
        if rc == -1:          # TN_ERR_INVALID
            raise UnsupportedJpeg("libtennis_hip tn_jpeg_decode refused the input: " + self.ctx.lib.tn_last_error().decode("utf-8", "replace"))
        check(rc, "tn_jpeg_decode")
        return out

    @property
    def sync_passes(self) -> int:
        return int(self.ctx.lib.tn_jpeg_sync_passes(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None:
            self.ctx.lib.tn_jpeg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def _decoder(ctx=None) -> JpegDecoder:
    ctx = ctx if ctx is not None else _lib.default_context()
    d = _default.get(id(ctx))
    if d is None:
        d = _default[id(ctx)] = JpegDecoder(ctx)
    return d


def imdecode_batch(bufs, ctx=None) -> torch.Tensor:
    return _decoder(ctx).decode(bufs)


def imread_batch(paths, ctx=None) -> torch.Tensor:
    bufs = []
    for p in paths:
        with open(p, "rb") as f:
            bufs.append(f.read())
    return imdecode_batch(bufs, ctx)


def imdecode(buf, flag=1, to_rgb=1, ctx=None) -> torch.Tensor:
    """``mx.image.imdecode(buf, flag=1, to_rgb=1)``: one image, (H, W, 3) uint8 RGB on the GPU."""
    if flag != 1 or to_rgb != 1:
        raise NotImplementedError("only flag=1, to_rgb=1 (3-channel RGB), as the reference calls it (dataset.py:204)")
    return imdecode_batch([buf], ctx)[0]


def imread(path, flag=1, to_rgb=1, ctx=None) -> torch.Tensor:
    """``mx.image.imread(path, 1)`` (reference dataset.py:204,216)."""
    with open(path, "rb") as f:
        return imdecode(f.read(), flag, to_rgb, ctx)
