"""ORACLE (test infrastructure only) — fp32 CPU restatement of the frame encoder.

PARITY UNPINNED: the reference's arithmetic for this path lives in un-pinned
third-party packages that are absent here (mxnet 1.x, gluoncv model_zoo —
reference evaluate.py:16,125; train.py:18,204) and the reference has no tests
or golden vectors.  This file restates the *published*
DenseNet-121 ``.features`` graph (Huang et al. 2017; GluonCV
``model_zoo/densenet.py`` [EXT, SURVEY App. A/B]) and is cross-checked against
``torch.nn.functional`` on CPU in tests/test_cpu_oracle.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product path
(``tennis_amd``) never does.

Conventions restated (all [EXT]):
  * convs have no bias; dense layer = concat(input, new) with the input first;
  * BatchNorm inference: y = gamma*(x-mean)/sqrt(var+1e-5)+beta;
  * stem maxpool 3x3/2 pad 1 pads with -inf; transition AvgPool 2x2/2 no pad;
  * ``.features`` ends with BN, ReLU, AvgPool2D(7) (stride 7, floor), Flatten
    in NCHW order -> 1024-d at 224^2, 4096-d at 512^2 (reference train.py:259).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5


def conv2d_nhwc(x: np.ndarray, w: np.ndarray, stride: int, pad: int) -> np.ndarray:
    """x (B,H,W,C) fp32, w Gluon layout (O,C,kh,kw) -> (B,Ho,Wo,O); im2col + sgemm."""
    b, h, wd, c = x.shape
    o, ci, kh, kw = w.shape
    assert ci == c
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    if kh == 1 and kw == 1 and stride == 1 and pad == 0:
        return (x.reshape(-1, c) @ w.reshape(o, c).T).reshape(b, h, wd, o)
    xp = np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0))) if pad else x
    out = np.zeros((b, ho, wo, o), dtype=np.float32)
    # accumulate tap by tap: keeps the temporary at one (B*Ho*Wo, C) slab
    for ky in range(kh):
        for kx in range(kw):
            sl = xp[:, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride, :]
            out += (sl.reshape(-1, c) @ w[:, :, ky, kx].T).reshape(b, ho, wo, o)
    return out


def batchnorm(x, p, name):
    s = p[name + "_gamma"] / np.sqrt(p[name + "_running_var"] + np.float32(BN_EPS))
    t = p[name + "_beta"] - p[name + "_running_mean"] * s
    return x * s.astype(np.float32) + t.astype(np.float32)


def relu(x):
    return np.maximum(x, np.float32(0))


def maxpool3x3s2p1(x):
    b, h, w, c = x.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    xp = np.full((b, h + 2, w + 2, c), -np.inf, dtype=np.float32)
    xp[:, 1:-1, 1:-1, :] = x
    out = np.full((b, ho, wo, c), -np.inf, dtype=np.float32)
    for ky in range(3):
        for kx in range(3):
            out = np.maximum(out, xp[:, ky:ky + 2 * (ho - 1) + 1:2, kx:kx + 2 * (wo - 1) + 1:2, :])
    return out


def avgpool(x, k):
    """AvgPool2D(k) with stride k, no padding, floor ('valid')."""
    b, h, w, c = x.shape
    ho, wo = h // k, w // k
    return x[:, :ho * k, :wo * k, :].reshape(b, ho, k, wo, k, c).mean(axis=(2, 4), dtype=np.float32)


def densenet121_features(x_nchw: np.ndarray, p: dict, prefix: str = "densenet0_", taps: dict | None = None):
    """``get_model('DenseNet121').features`` (call sites: reference evaluate.py:125,
    definitions.py:30).  x (B,3,H,W) fp32 -> (B,F) fp32; optional ``taps`` dict
    receives NHWC stage outputs for per-stage parity checks."""
    x = np.ascontiguousarray(x_nchw.transpose(0, 2, 3, 1)).astype(np.float32)
    x = conv2d_nhwc(x, p[prefix + "conv0_weight"], 2, 3)
    x = relu(batchnorm(x, p, prefix + "batchnorm0"))
    if taps is not None:
        taps["stem"] = x
    x = maxpool3x3s2p1(x)
    if taps is not None:
        taps["pool0"] = x
    outer = 1
    for st, nl in enumerate((6, 12, 24, 16), start=1):
        for li in range(nl):
            sp = f"{prefix}stage{st}_"
            y = relu(batchnorm(x, p, f"{sp}batchnorm{2 * li}"))
            y = conv2d_nhwc(y, p[f"{sp}conv{2 * li}_weight"], 1, 0)
            if taps is not None and li == 0:
                taps[f"stage{st}_l0_bottleneck"] = y
            y = relu(batchnorm(y, p, f"{sp}batchnorm{2 * li + 1}"))
            y = conv2d_nhwc(y, p[f"{sp}conv{2 * li + 1}_weight"], 1, 1)
            x = np.concatenate([x, y], axis=-1)
        if taps is not None:
            taps[f"stage{st}"] = x
        if st != 4:
            y = relu(batchnorm(x, p, f"{prefix}batchnorm{outer}"))
            y = conv2d_nhwc(y, p[f"{prefix}conv{outer}_weight"], 1, 0)
            x = avgpool(y, 2)
            if taps is not None:
                taps[f"trans{st}"] = x
            outer += 1
    x = relu(batchnorm(x, p, f"{prefix}batchnorm{outer}"))
    x = avgpool(x, 7)                                   # (B, H', W', 1024)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).reshape(x.shape[0], -1)  # NCHW flatten


def dense(x, p, prefix):
    """``nn.Dense(units, flatten=True)`` (reference definitions.py:25)."""
    return x.reshape(x.shape[0], -1) @ p[prefix + "weight"].T + p[prefix + "bias"]
