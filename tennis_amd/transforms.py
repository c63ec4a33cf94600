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

Round 4: the reference's TRAIN transform (train.py:127-139) is recognised too:

    transform_train = transforms.Compose([transforms.RandomResizedCrop(data_shape), transforms.RandomFlipLeftRight(),
                                          transforms.RandomColorJitter(brightness=0.4, contrast=0.4, saturation=0.4),
                                          transforms.RandomLighting(0.1), transforms.ToTensor(), transforms.Normalize(mean, std)])

The random parameters of a batch are drawn on the host the way ``mx.image.random_size_crop`` and the ``image_random`` operators
draw them (ten attempts at an area in ``scale`` and a log-uniform aspect ratio, then the centre crop; alphas ``1 + U(-p, p)``,
a random order of the jitter operators, AlexNet's PCA lighting) from a seeded numpy generator - MXNet's own random stream
cannot be reproduced - and the image arithmetic runs in three HIP launches per batch (``tn_augment_forward``), uint8 in, uint8 out.
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


class RandomResizedCrop:
    """``transforms.RandomResizedCrop(size, scale=(0.08, 1.0), ratio=(3/4, 4/3), interpolation=1)`` (reference train.py:130)."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), interpolation=1):
        if interpolation != 1 or not isinstance(size, int):
            raise NotImplementedError("only RandomResizedCrop(int size, interpolation=1) (train.py:130)")
        self.size, self.scale, self.ratio = size, tuple(scale), tuple(ratio)

    def draw(self, rng, h, w):
        """-> (x0, y0, cw, ch): mx.image.random_size_crop [EXT]."""
        area = h * w
        for _ in range(10):
            target = rng.uniform(self.scale[0], self.scale[1]) * area
            r = np.exp(rng.uniform(np.log(self.ratio[0]), np.log(self.ratio[1])))
            cw, ch = int(round(np.sqrt(target * r))), int(round(np.sqrt(target / r)))
            if 0 < cw <= w and 0 < ch <= h:
                return int(rng.integers(0, w - cw + 1)), int(rng.integers(0, h - ch + 1)), cw, ch
        # mx.image.center_crop(src, (size, size)): the largest window of the target's aspect ratio that fits, centred
        cw, ch = self.size, self.size
        if h < ch:
            cw, ch = float(cw * h) / ch, h
        if w < cw:
            cw, ch = w, float(ch * w) / cw
        cw, ch = int(cw), int(ch)
        return int((w - cw) / 2), int((h - ch) / 2), cw, ch


class RandomFlipLeftRight:
    def draw(self, rng):
        return int(rng.random() < 0.5)


class RandomColorJitter:
    """``transforms.RandomColorJitter(brightness, contrast, saturation, hue=0)`` (reference train.py:132-133)."""

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0):
        if hue != 0:
            raise NotImplementedError("RandomColorJitter(hue != 0) is not used by the reference (train.py:132)")
        self.brightness, self.contrast, self.saturation = float(brightness), float(contrast), float(saturation)

    def draw(self, rng):
        """-> (order: the four operators 0 brightness / 1 contrast / 2 saturation / 3 hue in the order they run, alphas)"""
        order = [int(o) for o in rng.permutation(4)]
        a = [1.0 + rng.uniform(-p, p) if p > 0 else 1.0 for p in (self.brightness, self.contrast, self.saturation)]
        return order, a


class RandomLighting:
    """``transforms.RandomLighting(alpha)``: AlexNet-style PCA noise (reference train.py:134)."""
    EIGVAL = np.array([55.46, 4.794, 1.148], np.float32)
    EIGVEC = np.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140], [-0.5836, -0.6948, 0.4203]], np.float32)

    def __init__(self, alpha):
        self.alpha = float(alpha)

    def draw(self, rng):
        a = rng.normal(0.0, self.alpha, 3).astype(np.float32)
        return (self.EIGVEC @ (a * self.EIGVAL)).astype(np.float32)


class AugFrame(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("cw", C.c_int32), ("ch", C.c_int32), ("flip", C.c_int32), ("order", C.c_int32),
                ("brightness", C.c_float), ("contrast", C.c_float), ("saturation", C.c_float), ("light", C.c_float * 3)]


class Compose:
    device_batched = True      # TennisSet hands raw frames through; DataLoader calls this once per batch

    def __init__(self, transforms, ctx: _lib.Context | None = None, seed: int = 0):
        t = list(transforms)
        self.train = (len(t) == 6 and isinstance(t[0], RandomResizedCrop) and isinstance(t[1], RandomFlipLeftRight)
                      and isinstance(t[2], RandomColorJitter) and isinstance(t[3], RandomLighting) and isinstance(t[4], ToTensor)
                      and isinstance(t[5], Normalize))
        if self.train:
            if not (np.allclose(t[5].mean, IMAGENET_MEAN) and np.allclose(t[5].std, IMAGENET_STD)):
                raise NotImplementedError("the consumers' fused Normalize uses the ImageNet mean/std of train.py:138")
            self._aug = t[:4]
            self.crop = t[0].size
            self._ctx = ctx
            self._plans = {}
            self._rng = np.random.default_rng(seed)
            self.last_params = None          # the records of the last call (tests feed them to the oracle)
            return
        ok = (len(t) == 4 and isinstance(t[0], Resize) and isinstance(t[1], CenterCrop) and isinstance(t[2], ToTensor)
              and isinstance(t[3], Normalize))
        if not ok:
            raise NotImplementedError("Compose supports the reference's test transform [Resize(s), CenterCrop(c), ToTensor(), "
                                      "Normalize(mean, std)] and its train transform [RandomResizedCrop(s), RandomFlipLeftRight(), "
                                      "RandomColorJitter(b, c, s), RandomLighting(a), ToTensor(), Normalize(mean, std)]")
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

    def draw_params(self, n, h, w):
        """``n`` parameter records for frames of ``h x w`` pixels (one independent draw per frame, as the reference's per-sample
        transform call)."""
        rec = (AugFrame * n)()
        for i in range(n):
            x0, y0, cw, ch = self._aug[0].draw(self._rng, h, w)
            flip = self._aug[1].draw(self._rng)
            order, a = self._aug[2].draw(self._rng)
            light = self._aug[3].draw(self._rng)
            rec[i] = AugFrame(x0, y0, cw, ch, flip, sum(o << (2 * k) for k, o in enumerate(order)), a[0], a[1], a[2],
                              (C.c_float * 3)(*[float(v) for v in light]))
        return rec

    def augment(self, x, rec):
        """(N, H, W, 3) uint8 device frames + ``N`` records -> (N, size, size, 3) uint8 on the device"""
        if self._ctx is None:
            self._ctx = _lib.default_context()
        n, h, w = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        s = self.crop
        dev = x.device
        out = torch.empty((n, s, s, 3), dtype=torch.uint8, device=dev)
        tmp = torch.empty((n, s, s, 3), dtype=torch.uint8, device=dev)
        gray = torch.empty((n,), dtype=torch.float32, device=dev)
        rec_dev = torch.from_numpy(np.frombuffer(bytes(rec), np.uint8).copy()).to(dev)
        check(self._ctx.lib.tn_augment_forward(self._ctx.handle, ptr(x), n, h, w, ptr(rec_dev), C.cast(rec, C.c_void_p), s, ptr(tmp),
                                               ptr(gray), ptr(out)), "tn_augment_forward")
        return out

    def __call__(self, frames):
        x = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
        if x.dtype != torch.uint8 or x.shape[-1] != 3 or x.dim() not in (3, 4, 5):
            raise ValueError(f"expected decoded uint8 RGB frames (..., H, W, 3), got {tuple(x.shape)} {x.dtype}")
        lead = tuple(x.shape[:-3])
        h, w = int(x.shape[-3]), int(x.shape[-2])
        if self.train:
            if self._ctx is None:
                self._ctx = _lib.default_context()
            x = x.to(torch.device('cuda', self._ctx.device)).contiguous().reshape(-1, h, w, 3)
            out = torch.empty((x.shape[0], self.crop, self.crop, 3), dtype=torch.uint8, device=x.device)
            recs = []
            for s in range(0, x.shape[0], 65535):
                n = min(65535, x.shape[0] - s)
                rec = self.draw_params(n, h, w)
                recs.append(rec)
                out[s:s + n] = self.augment(x[s:s + n], rec)
            self.last_params = recs
            return out.reshape(lead + (self.crop, self.crop, 3))
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
