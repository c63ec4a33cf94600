"""Calibrated fp16 conversion of a trained fp32 DenseNet-121 checkpoint (round 3).

``north_star`` asks for features / logits "within 1e-3 of the MXNet CPU reference", which evaluates the fp32 parameters
(reference models/vision/definitions.py:27-33).  Plain rounding of the 6.9 M conv weights to fp16 costs 3.2e-3 on the pooled
features; the exact-weights mode (hi + lo fp16 pairs) keeps 1e-3 at twice the MFMAs.  This module gets there with ONE fp16
number per weight: a handful of calibration frames go through the encoder's layer-wise kernels
(``tn_densenet121_input_means``), and ``weights.as_fp16_model(params, input_means=...)`` then picks, weight by weight, the fp16
neighbour that keeps the mean-activation-weighted rounding error of each output row at zero."""
from __future__ import annotations

import torch

from . import weights as W


def calibrated_fp16_model(params: dict, frames: torch.Tensor, size=224, prefix: str = "densenet0_", ctx=None) -> dict:
    """``params``: fp32 parameters (Gluon names); ``frames``: calibration frames on the GPU in any layout the encoder takes
    (NHWC u8 / NHWC fp16 normalised / NCHW fp32 normalised).  Returns the converted parameter dict (conv weights
    fp16-representable after the BN2 fold, everything else untouched) for ``DenseNet121Features`` / ``get_model``."""
    from .engine import DenseNet121Features
    plain = W.as_fp16_model(params)                  # the statistics barely depend on how the weights were rounded
    enc = DenseNet121Features(plain, size, max_batch=int(frames.shape[0]), prefix=prefix, ctx=ctx)
    means = enc.input_means(frames, prefix=prefix)
    del enc
    return W.as_fp16_model(params, input_means=means)
