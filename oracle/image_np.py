"""ORACLE (test infrastructure only) — CPU restatement of the geometric half of the reference's test transform.

  transforms.Resize(data_shape + 32)   reference evaluate.py:94 -> gluon Resize(size, keep_ratio=False, interpolation=1)
                                       -> mx.image.imresize(src, w, h, interp=1) -> cv::resize(INTER_LINEAR) on uint8 [EXT]
  transforms.CenterCrop(data_shape)    reference evaluate.py:95 -> mx.image.center_crop: x0 = int((w - new_w) / 2) [EXT]
  ToTensor + Normalize                 reference evaluate.py:96-97 (restated in tennis_amd.dataset.default_transform's
                                       formula: x / 255, (x - mean) / std, HWC -> CHW)

PARITY UNPINNED: OpenCV and MXNet are absent.  The resize follows OpenCV's published generic 8-bit bilinear
(imgproc/resize.cpp: coordinates (d + 0.5) * scale - 0.5 in float, INTER_RESIZE_COEF_BITS = 11 coefficients rounded
half-to-even, HResizeLinear in int32, VResizeLinear's (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2,
and the INTER_AREA 2x2 box average substituted for an exact 2x reduction); an IPP-enabled OpenCV build may differ by
one grey level.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


def _taps(src, dst, clamp):
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * (float(src) / dst) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp:
        lo, hi = s < 0, s >= src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src - 1, s))
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)    # rint: half to even, as cvRound
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, c0, c1


def resize_bilinear_u8(img, out_h, out_w):
    """cv::resize(img, (out_w, out_h), INTER_LINEAR) for an (H, W, C) uint8 array."""
    H, W, _ = img.shape
    a = img.astype(np.int64)
    if W == 2 * out_w and H == 2 * out_h:                       # is_area_fast, iscale 2: 2x2 box
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, a0, a1 = _taps(W, out_w, True)
    sy, b0, b1 = _taps(H, out_h, False)
    x1 = np.minimum(sx + 1, W - 1)
    rows = a[:, sx] * a0[None, :, None] + a[:, x1] * a1[None, :, None]          # (H, out_w, C) int
    r0, r1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    S0, S1 = rows[r0], rows[r1]
    v = (((b0[:, None, None] * (S0 >> 4)) >> 16) + ((b1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


def center_crop(img, size):
    H, W, _ = img.shape
    x0, y0 = int((W - size) / 2), int((H - size) / 2)
    return img[y0:y0 + size, x0:x0 + size]


def test_transform_u8(img, data_shape):
    """Resize(data_shape + 32) + CenterCrop(data_shape): (H, W, 3) uint8 -> (data_shape, data_shape, 3) uint8."""
    s = data_shape + 32
    return center_crop(resize_bilinear_u8(img, s, s), data_shape)


# ---- the reference's train transform (train.py:127-136), image arithmetic only: the random parameters are inputs ----
#   transforms.RandomResizedCrop -> mx.image.random_size_crop -> fixed_crop (crop, then imresize interp=1)        [EXT]
#   transforms.RandomFlipLeftRight; RandomColorJitter -> the image_random operators (brightness / contrast / saturation, each
#   saturate_cast to uint8, in a drawn order; contrast against the image's mean grey 0.299 R + 0.587 G + 0.114 B);
#   RandomLighting -> in + eigvec (alpha * eigval)                                      [EXT: src/operator/image/image_random-inl.h]
# PARITY UNPINNED (MXNet absent): saturate_cast is taken as clamp-then-truncate, the image's mean grey as the exact mean of the
# per-pixel float32 grey values.

def _sat_u8(v):
    return np.clip(v, np.float32(0), np.float32(255)).astype(np.uint8)        # clamp, then truncate


def _gray(img):
    f = img.astype(np.float32)
    return (f[..., 0] * np.float32(0.299) + f[..., 1] * np.float32(0.587)) + f[..., 2] * np.float32(0.114)


def augment_u8(img, x0, y0, cw, ch, flip, order, brightness, contrast, saturation, light, size):
    """(H, W, 3) uint8 -> (size, size, 3) uint8; ``order``: the four jitter operators (0 brightness, 1 contrast, 2 saturation,
    3 hue = nothing) in the order they run; ``light``: the three per-channel offsets of RandomLighting."""
    a = resize_bilinear_u8(img[y0:y0 + ch, x0:x0 + cw], size, size)
    if flip:
        a = a[:, ::-1]
    b, c, s = np.float32(brightness), np.float32(contrast), np.float32(saturation)
    for op in order:
        f = a.astype(np.float32)
        if op == 0:
            a = _sat_u8(f * b)
        elif op == 1:
            gm = np.float32(_gray(a).astype(np.float64).sum() / (size * size))
            a = _sat_u8(f * c + (np.float32(1) - c) * gm)
        elif op == 2:
            g = _gray(a) * (np.float32(1) - s)
            a = _sat_u8(f * s + g[..., None])
    return _sat_u8(a.astype(np.float32) + np.asarray(light, np.float32)[None, None, :])
