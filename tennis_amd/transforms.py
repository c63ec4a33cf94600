"""``mxnet.gluon.data.vision.transforms`` surface of the reference's test transform (evaluate.py:93-98,
train.py's ``transform_test``), executed on the GPU:

    transform_test = transforms.Compose([transforms.Resize(data_shape + 32),
                                         transforms.CenterCrop(data_shape),
                                         transforms.ToTensor(),
                                         transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])

``Compose`` recognises exactly that chain.  Calling it on decoded uint8 RGB frames — one (H, W, 3) frame, a
(B, H, W, 3) batch or a (B, T, H, W, 3) window batch; numpy or torch — runs Resize + CenterCrop in one HIP launch
(``tn_preproc_*``, bit-exact OpenCV 8-bit bilinear) and returns the uint8 NHWC crop on the device; ToTensor +
Normalize are applied by the encoder's stem as it loads those bytes (``TN_LAYOUT_NHWC_U8``), so the result of the
chain is what the reference feeds its network, without the fp32 NCHW intermediate ever existing in memory.
There is no CPU implementation here: without the HIP library or a GPU the call raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class Resize:
    def __init__(self, size, keep_ratio=False, interpolation=1):
        if keep_ratio or interpolation != 1 or not isinstance(size, int):
            raise NotImplementedError("only Resize(int size, keep_ratio=False, interpolation=1) (evaluate.py:94)")
        self.size = size


class CenterCrop:
    def __init__(self, size, interpolation=1):
        if not isinstance(size, int):
            raise NotImplementedError("only CenterCrop(int size) (evaluate.py:95)")
        self.size = size


class ToTensor:
    pass


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = tuple(float(m) for m in mean), tuple(float(s) for s in std)


class Compose:
    device_batched = True      # TennisSet hands raw frames through; DataLoader calls this once per batch

    def __init__(self, transforms, ctx: _lib.Context | None = None):
        t = list(transforms)
        ok = (len(t) == 4 and isinstance(t[0], Resize) and isinstance(t[1], CenterCrop) and isinstance(t[2], ToTensor)
              and isinstance(t[3], Normalize))
        if not ok:
            raise NotImplementedError("Compose supports the reference's test transform only: "
                                      "[Resize(s), CenterCrop(c), ToTensor(), Normalize(mean, std)]")
        if not (np.allclose(t[3].mean, IMAGENET_MEAN) and np.allclose(t[3].std, IMAGENET_STD)):
            raise NotImplementedError("the stem's fused Normalize uses the ImageNet mean/std of evaluate.py:97")
        if t[1].size > t[0].size:
            raise NotImplementedError("CenterCrop larger than the resized frame")
        self.resize, self.crop = t[0].size, t[1].size
        self._ctx = ctx
        self._plans = {}          # (src_h, src_w) -> tn_preproc handle

    def _plan(self, h, w):
        if (h, w) not in self._plans:
            if self._ctx is None:
                self._ctx = _lib.default_context()
            hd = C.c_void_p()
            check(self._ctx.lib.tn_preproc_create(self._ctx.handle, h, w, self.resize, self.crop, C.byref(hd)),
                  "tn_preproc_create")
            self._plans[(h, w)] = hd
        return self._plans[(h, w)]

    def __call__(self, frames):
        x = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
        if x.dtype != torch.uint8 or x.shape[-1] != 3 or x.dim() not in (3, 4, 5):
            raise ValueError(f"expected decoded uint8 RGB frames (..., H, W, 3), got {tuple(x.shape)} {x.dtype}")
        lead = tuple(x.shape[:-3])
        h, w = int(x.shape[-3]), int(x.shape[-2])
        plan = self._plan(h, w)
        x = x.to(torch.device('cuda', self._ctx.device)).contiguous().reshape(-1, h, w, 3)
        out = torch.empty((x.shape[0], self.crop, self.crop, 3), dtype=torch.uint8, device=x.device)
        for s in range(0, x.shape[0], 65535):
            n = min(65535, x.shape[0] - s)
            check(self._ctx.lib.tn_preproc_forward(plan, ptr(x[s:s + n]), n, ptr(out[s:s + n])), "tn_preproc_forward")
        return out.reshape(lead + (self.crop, self.crop, 3))

    def __del__(self):
        try:
            for hd in self._plans.values():
                self._ctx.lib.tn_preproc_destroy(hd)
            self._plans = {}
        except Exception:
            pass
