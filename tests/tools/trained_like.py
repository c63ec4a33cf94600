"""(test infrastructure: imports oracle/)  A "trained-looking" DenseNet-121 parameter set.

`weights.make_densenet121_weights` draws He-normal filters and BatchNorm gammas in [0.8, 1.2]: every channel matters equally and
no scale is extreme.  A trained checkpoint does not look like that, and the calibrated fp16 conversion (bias-corrected vector
feedback, BN1 as a clamp with its scale folded into the 1x1 weights) is exactly the kind of code that breaks on what it has not
seen (a saturating fold on tiny scales).  Here:
  * conv filters are CORRELATED: rows share low-rank components, 3x3 / 7x7 kernels are spatially smooth blends, and every
    output row has a log-normal gain (heavy-tailed row norms);
  * BatchNorm gammas are log-normal with ~7 % near-dead channels (1e-2 ... 1e-6 of the typical scale) and ~4 % negative ones;
    betas lean negative (sparser activations);
  * running_mean / running_var are the MEASURED statistics of the layer's input over a small mixed batch (as a trained
    network's are), jittered like an exponential moving average that lags - so activations stay O(1) through the 120
    convolutions whatever the filters do, and some channels have tiny or huge variances.
Seeded, deterministic, fp32, Gluon names.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from tennis_amd import calib_frames as CF
from tennis_amd import weights as W


LOWRANK, ROWGAIN, GAMMA_SIGMA, DEAD, NEG = 0.7, 0.45, 0.5, 0.07, 0.04      # knobs of the generator (module-level so that a study can vary them)


def _filters(rng, cout, cin, k):
    fan_in = cin * k * k
    rank = max(2, min(cin, cout) // 8)
    indep = rng.normal(size=(cout, cin, k, k))
    low = np.einsum("or,rikl->oikl", rng.normal(size=(cout, rank)), rng.normal(size=(rank, cin, k, k))) / np.sqrt(rank)
    w = 0.7 * indep + LOWRANK * low
    if k > 1:       # spatially smooth kernels: blend with a blurred copy
        t = torch.from_numpy(w.reshape(1, cout * cin, k, k))
        ker = torch.tensor([[1., 2, 1], [2, 4, 2], [1, 2, 1]], dtype=torch.float64) / 16
        blur = F.conv2d(t, ker.expand(cout * cin, 1, 3, 3).contiguous(), padding=1, groups=cout * cin).numpy().reshape(w.shape)
        w = 0.5 * w + 1.2 * blur
    w *= np.exp(rng.normal(0.0, ROWGAIN, (cout, 1, 1, 1)))                  # heavy-tailed row norms
    w *= np.sqrt(2.0 / fan_in) / w.std()
    return w.astype(np.float32)


def _bn_from_stats(rng, prefix, x, final=False):
    """BatchNorm parameters whose running statistics are those of ``x`` (N, C, H, W)"""
    c = x.shape[1]
    mean = x.mean((0, 2, 3)).numpy().astype(np.float64)
    var = x.var((0, 2, 3), unbiased=False).numpy().astype(np.float64)
    mean = mean + rng.normal(0, 0.05, c) * np.sqrt(var + 1e-12)              # a lagging moving average
    var = var * np.exp(rng.normal(0, 0.08, c))
    gamma = np.exp(rng.normal(np.log(1.0 if final else 0.6), 0.15 if final else GAMMA_SIGMA, c))
    if not final:
        dead = rng.random(c) < DEAD
        gamma[dead] *= 10.0 ** rng.uniform(-6, -2, int(dead.sum()))
        gamma[rng.random(c) < NEG] *= -1.0
    beta = rng.normal(0.1 if final else -0.15, 0.25, c)
    return {prefix + "_gamma": gamma.astype(np.float32), prefix + "_beta": beta.astype(np.float32),
            prefix + "_running_mean": mean.astype(np.float32), prefix + "_running_var": np.maximum(var, 1e-10).astype(np.float32)}


@torch.no_grad()
def make_trained_like_weights(seed: int = 0, prefix: str = "densenet0_", size: int = 224, stat_frames: int = 36) -> dict:
    rng = np.random.default_rng([seed, 4242])
    frames, _ = CF.mixed_batch(stat_frames, size, seed=seed + 31, fine=stat_frames // 9)
    x = torch.from_numpy(W.normalize_to_nchw_f32(frames))
    p = {}

    def bnrelu(x, name, final=False):
        p.update(_bn_from_stats(rng, prefix + name, x, final))
        q = {k: torch.from_numpy(p[prefix + name + s]) for k, s in (("m", "_running_mean"), ("v", "_running_var"), ("g", "_gamma"), ("b", "_beta"))}
        return F.relu(F.batch_norm(x, q["m"], q["v"], q["g"], q["b"], False, 0.0, W.BN_EPS))

    def conv(x, name, cout, k, **kw):
        w = _filters(rng, cout, x.shape[1], k)
        p[prefix + name + "_weight"] = w
        return F.conv2d(x, torch.from_numpy(w), **kw)

    x = conv(x, "conv0", W.INIT_FEATURES, 7, stride=2, padding=3)
    x = F.max_pool2d(bnrelu(x, "batchnorm0"), 3, 2, 1)
    outer = 1
    for st, nl in enumerate(W.BLOCK_CONFIG, 1):
        for li in range(nl):
            y = conv(bnrelu(x, f"stage{st}_batchnorm{2 * li}"), f"stage{st}_conv{2 * li}", W.BN_SIZE * W.GROWTH, 1)
            y = conv(bnrelu(y, f"stage{st}_batchnorm{2 * li + 1}"), f"stage{st}_conv{2 * li + 1}", W.GROWTH, 3, padding=1)
            x = torch.cat([x, y], 1)
        if st != len(W.BLOCK_CONFIG):
            x = conv(bnrelu(x, f"batchnorm{outer}"), f"conv{outer}", x.shape[1] // 2, 1)
            x = F.avg_pool2d(x, 2, 2)
            outer += 1
    bnrelu(x, f"batchnorm{outer}", final=True)
    return p
