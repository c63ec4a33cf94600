"""Seeded synthetic parameters for the hot path, keyed by Gluon parameter names.

The reference never ships weights in-tree: the frame encoder comes from
``gluoncv.model_zoo.get_model('DenseNet121', pretrained=True).features``
(reference evaluate.py:125, train.py:204) and trained checkpoints live on
Google Drive (reference models/README.md:2).  With no network, every test,
the oracle and the bench use the seeded tensors made here.  Keys follow the
Gluon naming convention [EXT, SURVEY App. B] so that a ``.params`` reader can
drop real checkpoints in later (SURVEY §8b "Weights format").

Shapes follow Gluon: conv ``(C_out, C_in, kh, kw)``, dense ``(units, in)``,
RNN ``{l,r}0_i2h_weight (G*H, F)``, ``{l,r}0_h2h_weight (G*H, H)``.

Statistics are chosen so that activations stay O(1) through all 120
convolutions: He-normal conv weights; BN gamma near 1,
beta near 0, running_mean near 0 and running_var near the analytic variance of
the producing op.
"""
from __future__ import annotations

import numpy as np

GROWTH = 32
BN_SIZE = 4
INIT_FEATURES = 64
BLOCK_CONFIG = (6, 12, 24, 16)
BN_EPS = 1e-5


def densenet121_layout():
    """Static description of DenseNet-121 ``.features``.

    Returns a list of dicts, one per conv, in execution order:
    ``{"name", "kind": stem|dense1x1|dense3x3|trans, "cin", "cout", "stage", "layer"}``
    plus the matching BN name in ``"bn"`` (the BN that precedes the conv, or for
    the stem the BN that follows it).
    """
    convs = [dict(name="conv0", bn="batchnorm0", kind="stem", cin=3, cout=INIT_FEATURES, stage=0, layer=0)]
    c = INIT_FEATURES
    outer = 1
    for si, nl in enumerate(BLOCK_CONFIG):
        st = si + 1
        for li in range(nl):
            convs.append(dict(name=f"stage{st}_conv{2 * li}", bn=f"stage{st}_batchnorm{2 * li}",
                              kind="dense1x1", cin=c + GROWTH * li, cout=BN_SIZE * GROWTH, stage=st, layer=li))
            convs.append(dict(name=f"stage{st}_conv{2 * li + 1}", bn=f"stage{st}_batchnorm{2 * li + 1}",
                              kind="dense3x3", cin=BN_SIZE * GROWTH, cout=GROWTH, stage=st, layer=li))
        c += GROWTH * nl
        if si != len(BLOCK_CONFIG) - 1:
            convs.append(dict(name=f"conv{outer}", bn=f"batchnorm{outer}", kind="trans",
                              cin=c, cout=c // 2, stage=st, layer=0))
            c //= 2
            outer += 1
    return convs, f"batchnorm{outer}", c  # final BN name, final channel count (1024)


def _bn(rng, prefix, c, var_center=1.0):
    return {
        prefix + "_gamma": rng.uniform(0.8, 1.2, c).astype(np.float32),
        prefix + "_beta": rng.normal(0.0, 0.1, c).astype(np.float32),
        prefix + "_running_mean": rng.normal(0.0, 0.1, c).astype(np.float32),
        prefix + "_running_var": (var_center * rng.uniform(0.8, 1.2, c)).astype(np.float32),
    }


BN_EPS = 1e-5     # gluon BatchNorm default (csrc/api.hip kBnEps)


def _round_fp16_error_feedback(wf: np.ndarray, mean: np.ndarray) -> np.ndarray:
    """(N, K) float64 values -> fp16-representable values: each entry is one of its two fp16 neighbours, chosen walking k in
    order so that the running sum over k of ``(rounded - exact) * mean[k]`` stays as close to zero as it can."""
    rtn = wf.astype(np.float16)
    rtn_f = rtn.astype(np.float64)
    other = np.nextafter(rtn, np.where(wf > rtn_f, np.float16(np.inf), np.float16(-np.inf)).astype(np.float16)).astype(np.float64)
    other = np.where(np.isfinite(other), other, rtn_f)
    e1, e2 = (rtn_f - wf) * mean[None, :], (other - wf) * mean[None, :]
    out = rtn_f.copy()
    acc = np.zeros(wf.shape[0])
    for k in range(wf.shape[1]):
        use2 = np.abs(acc + e2[:, k]) < np.abs(acc + e1[:, k])
        out[use2, k] = other[use2, k]
        acc += np.where(use2, e2[:, k], e1[:, k])
    return out


_VEC_RIDGE, _VEC_SWEEPS = 0.05, 3
_USE_LIBRARY_ROUNDING = True      # (scripts/calib_study.py and the tests also run the numpy reference)


def _round_fp16_vector_feedback(wf: np.ndarray, A: np.ndarray, sweeps: int | None = None, ridge: float | None = None,
                                use_library: bool = True) -> np.ndarray:
    """(N, K) float64 values -> fp16-representable values, each one of its two fp16 neighbours, chosen so that the rounding
    error of every output row is (nearly) orthogonal to ALL rows of ``A`` (F, K) at once - the mean input activations of F
    calibration frames - instead of to their average only (``_round_fp16_error_feedback``): minimises, per output row n,
    ``|| A d_n ||^2 / F + ridge * sum_k (d_n[k] * a_rms[k])^2`` over the 2^K neighbour choices by one greedy pass along k followed
    by ``sweeps`` passes of coordinate descent (each weight re-decided against the residual of all the others).  The ridge term
    keeps a weight on its nearest neighbour unless moving it buys something, which bounds what the conversion can do to frames
    whose activations lie outside the span of the calibration set (they see at most the plain-rounding error statistics).

    The work is done by the library's host routine ``tn_round_fp16_calibrated`` (csrc/calib_host.hip; no GPU involved); the numpy
    code below is the reference it is tested against (``use_library=False``)."""
    sweeps = _VEC_SWEEPS if sweeps is None else sweeps
    ridge = _VEC_RIDGE if ridge is None else ridge
    N, K = wf.shape
    F = A.shape[0]
    if use_library:
        import ctypes as C
        from . import _lib
        w32 = np.ascontiguousarray(wf, np.float32)
        A64 = np.ascontiguousarray(A, np.float64)
        out = np.empty((N, K), np.float32)
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        _lib.check(_lib.load().tn_round_fp16_calibrated(vp(w32), N, K, vp(A64), F, int(sweeps), float(ridge), vp(out)), "tn_round_fp16_calibrated")
        return out.astype(np.float64)
    wf = wf.astype(np.float32).astype(np.float64)            # (the library sees fp32 weights)
    rtn = wf.astype(np.float16)
    rtn_f = rtn.astype(np.float64)
    other = np.nextafter(rtn, np.where(wf > rtn_f, np.float16(np.inf), np.float16(-np.inf)).astype(np.float16)).astype(np.float64)
    other = np.where(np.isfinite(other) & (wf != rtn_f), other, rtn_f)
    d1, d2 = rtn_f - wf, other - wf                          # the two possible errors of each weight (N, K)
    An = A / np.sqrt(F)                                       # mean square over the frames
    a2 = (An * An).sum(0)                                     # (K,) = a_rms^2
    use2 = np.zeros((N, K), bool)
    r = np.zeros((N, F))
    for sweep in range(sweeps + 1):
        for k in range(K):
            ak = An[:, k]
            if sweep:
                r -= np.where(use2[:, k], d2[:, k], d1[:, k])[:, None] * ak[None, :]
            # cost of choice c: || r + d_c a_k ||^2 + ridge d_c^2 a2_k  =  const + 2 d_c (r . a_k) + d_c^2 a2_k (1 + ridge)
            ra = r @ ak
            c1 = 2 * d1[:, k] * ra + d1[:, k] ** 2 * a2[k] * (1 + ridge)
            c2 = 2 * d2[:, k] * ra + d2[:, k] ** 2 * a2[k] * (1 + ridge)
            u = c2 < c1
            use2[:, k] = u
            r += np.where(u, d2[:, k], d1[:, k])[:, None] * ak[None, :]
    return np.where(use2, other, rtn_f)


def bn_relu_clamp_fold(params: dict, bn_name: str, use_library: bool = False):
    """``(lo, hi, sw, tc)`` of BatchNorm ``bn_name`` followed by ReLU in the fused dense layers' rounding-free form
    ``relu(scale x + shift) = sw clamp(x, lo, hi) + tc`` (csrc/calib_host.hip::bn_relu_clamp_fold): ``lo`` / ``hi`` fp16 numbers -
    the ReLU threshold ``-shift / scale`` on the side the scale's sign says, +-65504 on the other - ``sw`` the factor that goes
    into column k of the 1x1 weights before they are rounded, ``tc`` the constant whose weighted sum joins the next BatchNorm's
    shift.  The numpy code is the reference; ``use_library`` runs the library's own host routine (``tn_bn_relu_clamp_fold``, no GPU
    involved) - tests/test_cpu_oracle.py holds the two together bit for bit."""
    g, b, mu, var = (np.ascontiguousarray(params[bn_name + sfx], np.float32) for sfx in ("_gamma", "_beta", "_running_mean", "_running_var"))
    n = g.size
    if use_library:
        import ctypes as C
        from . import _lib
        lo, hi, sw, tc = (np.empty(n, np.float32) for _ in range(4))
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        _lib.check(_lib.load().tn_bn_relu_clamp_fold(vp(g), vp(b), vp(mu), vp(var), n, vp(lo), vp(hi), vp(sw), vp(tc)), "tn_bn_relu_clamp_fold")
        return lo, hi, sw, tc
    with np.errstate(all="ignore"):
        scale = (g / np.sqrt(var + np.float32(BN_EPS))).astype(np.float32)
        shift = (b - (mu * scale).astype(np.float32)).astype(np.float32)
        s = scale.astype(np.float64)
        t = np.where(np.isfinite(shift), shift, 0).astype(np.float64)
        c = -t / np.where(s != 0, s, 1.0)
        kmax = 65504.0
        const = ~np.isfinite(scale) | (s == 0)
        dead = (~const) & (((s > 0) & (c > kmax)) | ((s < 0) & (c < -kmax)))          # always on the clipped side
        c16 = np.clip(c, -kmax, kmax).astype(np.float32).astype(np.float16).astype(np.float32)
        lo = np.where(s > 0, c16, np.float32(-kmax)).astype(np.float32)
        hi = np.where(s > 0, np.float32(kmax), c16).astype(np.float32)
        sw, tc = scale.copy(), t.astype(np.float32)
        off = const | dead
        lo[off] = 0; hi[off] = 0; sw[off] = 0
        tc[dead] = 0
        tc[const] = np.maximum(t[const], 0).astype(np.float32)
    return lo, hi, sw, tc


def _apply_bias_correction(out: dict, bias: dict, prefix: str):
    """Round 5.  ``bias[conv weight name]`` = the mean error of that convolution's output channels under the converted weights,
    ``sum_k (w_converted - w)[n, k] E[a_k]`` over the calibration frames and their pixels.  It is a constant per channel, and every
    consumer of the channel is a BatchNorm: the constant is added to the ``running_mean`` of each of them (the BatchNorm behind a 1x1
    / the stem; for the 32 new channels of a dense layer and for a transition's outputs every later BatchNorm of the block that
    reads them, and the block's closing BatchNorm), which removes it exactly, in front of the ReLU.  The rounding freedom of the
    calibrated conversion is then spent on the frame-to-frame VARIATION of the channel means only (``as_fp16_model`` centres
    its constraints).  Free at run time: same kernels, same parameter count.  [post-training-quantisation "bias correction"]"""
    cfg = BLOCK_CONFIG
    cin = [INIT_FEATURES]
    for b in range(len(cfg) - 1):
        cin.append((cin[b] + GROWTH * cfg[b]) // 2)

    def shift(bn, lo, b):
        k = prefix + bn + "_running_mean"
        if k not in out or b is None:
            return
        a = np.array(out[k], np.float32, copy=True)
        a[lo:lo + b.size] = (a[lo:lo + b.size].astype(np.float64) + b).astype(np.float32)
        out[k] = a

    shift("batchnorm0", 0, bias.get(prefix + "conv0_weight"))
    for st in range(1, len(cfg) + 1):
        nl = cfg[st - 1]
        term = f"batchnorm{st}"            # the BatchNorm behind the block: transition st, or the head's
        if st > 1:                         # the transition in front of this block feeds channels [0, cin)
            b = bias.get(prefix + f"conv{st - 1}_weight")
            for l in range(nl):
                shift(f"stage{st}_batchnorm{2 * l}", 0, b)
            shift(term, 0, b)
        for l in range(nl):
            shift(f"stage{st}_batchnorm{2 * l + 1}", 0, bias.get(prefix + f"stage{st}_conv{2 * l}_weight"))
            b = bias.get(prefix + f"stage{st}_conv{2 * l + 1}_weight")
            c0 = cin[st - 1] + GROWTH * l
            for l2 in range(l + 1, nl):
                shift(f"stage{st}_batchnorm{2 * l2}", c0, b)
            shift(term, c0, b)


def as_fp16_model(params: dict, input_means: dict | None = None, bias_correction: bool = True) -> dict:
    """Model conversion for the fp16 encoder: every conv ``*_weight`` is rounded once to
    fp16 (kept as fp32 arrays).  The served model IS these converted weights — the GPU
    path and the fp32 CPU oracle both evaluate them — so weight quantisation is a one-off
    conversion step, not kernel error.  BN statistics, Dense and RNN parameters stay fp32.

    The 1x1 convolution of a dense layer (``stageB_conv{2l}``) is followed directly by a
    BatchNorm (``stageB_batchnorm{2l+1}``) whose scale the encoder folds into the
    weights (the usual conv-BN fusion; csrc/api.hip, csrc/dense_strip_impl.h): for those the
    number that is rounded to fp16 is ``scale[n] * w[n][k]``, and the converted weight is
    ``fp16(scale[n] w[n][k]) / scale[n]`` — one rounding per weight either way.  Round 5: the factor also carries the SCALE
    ``sw[k]`` of the BatchNorm in front of the convolution, whose ReLU the kernels evaluate as a clamp of the stored activation
    (``bn_relu_clamp_fold``): the number rounded is ``scale[n] sw[k] w[n][k]`` (a channel with ``sw[k] = 0`` - a constant -
    keeps its weight as it is; the kernels never multiply it).

    (Measured on MI355X: with un-rounded fp32 conv weights the pooled features differ by
    up to 3.3e-3 because weight rounding is coherent across the 49 pooled pixels; with
    converted weights the kernels' own error is 7e-4 max, DESIGN.md "Numerics".)

    ``input_means`` values may be 2-D, one row per calibration FRAME (round 4, what ``tennis_amd.calibrate`` passes): the rounding
    error of every row is then made orthogonal to every frame's mean activations at once (``_round_fp16_vector_feedback``) -
    a conversion calibrated on the average only falls back towards plain rounding on frames whose channel means differ from
    the calibration set's (measured: DESIGN.md "Numerics", tests/test_gpu_calibration.py).

    ``input_means`` (round 3; ``engine.DenseNet121Features.input_means`` / ``tennis_amd.calibrate``): CALIBRATED rounding for
    parameters that are NOT fp16-representable (a trained fp32 checkpoint).  Round-to-nearest errors of a weight row are
    independent, but they all multiply activations that are positive behind a ReLU: the part of the row's error that survives the
    average pool is ``sum_k (w16[k] - w[k]) * E[a[k]]``.  With the mean activation of every input channel known, each weight is
    rounded to whichever of its two fp16 neighbours keeps that running sum closest to zero (error feedback along k): still ONE
    fp16 number per weight, same kernels, same speed - and the conversion error of the pooled features drops from 3.2e-3 to
    4e-4, which is what the hi + lo weight pairs of the exact-weights mode bought at twice the MFMAs.

    Round 5, ``bias_correction`` (with 2-D ``input_means`` only): the MEAN of every convolution's conversion error over the
    calibration frames is a per-channel constant and goes into the consuming BatchNorms' running means
    (``_apply_bias_correction``); the vector feedback works on the frames' deviations from that mean.  Measured on the CPU graph
    (scripts/conv_study.py, 15 families): conversion error rms 1.02e-4 -> 0.69e-4, worst 9.8e-4 -> 7.2e-4."""
    import re
    out = dict(params)
    bias, prefix = {}, None
    # The vector feedback is centred (its constraints are the frames' DEVIATIONS from the mean operand) only if the mean error it then
    # ignores is really taken out by _apply_bias_correction - which needs the whole DenseNet-121 tree under one prefix (a
    # partial dict used to get the centring without the correction)
    stem_keys = [k for k in params if k.endswith("conv0_weight") and getattr(params[k], "shape", (0,))[1:] == (3, 7, 7)]
    can_correct = bool(bias_correction and stem_keys and all(
        (stem_keys[0][:-len("conv0_weight")] + f"stage{st}_batchnorm{2 * l}_running_mean") in params for st in range(1, 5) for l in range(BLOCK_CONFIG[st - 1])))
    if bias_correction and input_means is not None and not can_correct and any(np.asarray(v).ndim == 2 for v in input_means.values()):
        import warnings
        warnings.warn("as_fp16_model: not a complete DenseNet-121 parameter tree - bias correction (and the centring of the calibration "
                      "constraints that goes with it) is off", RuntimeWarning, stacklevel=2)
    bias_correction = can_correct
    for k, v in params.items():
        if not (k.endswith("_weight") and v.ndim == 4):
            continue
        m = re.fullmatch(r"(.*stage\d+_)conv(\d+)_weight", k)
        bn = m and int(m.group(2)) % 2 == 0 and v.shape[2:] == (1, 1) and f"{m.group(1)}batchnorm{int(m.group(2)) + 1}"
        s = None
        if bn and bn + "_gamma" in params:
            s = (params[bn + "_gamma"] / np.sqrt(params[bn + "_running_var"] + np.float32(BN_EPS))).astype(np.float32)
            s = s.reshape(-1, 1, 1, 1)
            # round 5: the BatchNorm IN FRONT of the same convolution is evaluated as sw clamp(x, lo, hi) + tc
            # (csrc/calib_host.hip::bn_relu_clamp_fold); sw[k] multiplies column k of the weights before they are rounded
            bn1 = f"{m.group(1)}batchnorm{int(m.group(2))}"
            if bn1 + "_gamma" in params:
                s = s * bn_relu_clamp_fold(params, bn1)[2].reshape(1, -1, 1, 1)
        if s is None and k.endswith("conv0_weight") and v.shape[1:] == (3, 7, 7):
            # round 5: the stem works on x - 255 mean_c (an exact integer for uint8 frames), the input normalisation's
            # 1 / (255 std_c) is part of the weight that is rounded (csrc/common.h "the stem's operand")
            s = STEM_WFACTOR.reshape(1, 3, 1, 1)
            prefix = k[:-len("conv0_weight")]
        with np.errstate(over="ignore", invalid="ignore"):
            folded = (v * s).astype(np.float32) if s is not None else v.astype(np.float32)
        # the library refuses such a model at create ("a 1x1 weight leaves the fp16 range", csrc/api.hip): the conversion must not
        # hand out inf / NaN weights silently - nor let the oracle and a saved checkpoint carry them
        amax = float(np.abs(folded).max()) if folded.size else 0.0
        if not np.isfinite(amax) or amax > 65504.0:
            raise ValueError(f"as_fp16_model: {k} leaves the fp16 range once its BatchNorm scales are folded in (max |w| = {amax:.3g}); "
                             "serve this checkpoint with exact_weights=True")
        lost = int(((folded != 0) & (np.abs(folded) < 2.0 ** -25)).sum())
        if lost > max(8, folded.size // 1000):       # (a stray weight below 3e-8 is chance; a flushed column is a tiny BatchNorm scale)
            import warnings
            warnings.warn(f"as_fp16_model: {lost} non-zero weights of {k} round to 0 in fp16 (folded magnitude below 2^-25)", RuntimeWarning, stacklevel=2)
        if input_means is not None and k in input_means:
            am = np.asarray(input_means[k], np.float64)     # mean of the convolution's operand per input channel (per frame)
            taps = v.shape[2] * v.shape[3]
            if am.ndim == 2:      # one row per calibration frame: vector error feedback
                centred = am - am.mean(0, keepdims=True) if (bias_correction and am.shape[0] > 1) else am
                r = _round_fp16_vector_feedback(folded.astype(np.float64).reshape(v.shape[0], -1), np.repeat(centred, taps, axis=1),
                                                use_library=_USE_LIBRARY_ROUNDING)
            else:                # (cin, kh, kw) flattening
                r = _round_fp16_error_feedback(folded.astype(np.float64).reshape(v.shape[0], -1), np.repeat(am, taps))
            r = r.reshape(v.shape).astype(np.float32)
        else:
            r = folded.astype(np.float16).astype(np.float32)
        if s is not None:
            # Below fp16's normal range (a near-dead BatchNorm channel: scale 1e-5) the rounded value keeps only a few bits, and
            # un-folding it would hand the library - which also needs w[n][k] EXACTLY, for the constant tc[k] w[n][k] of the clamp
            # form - a weight that is up to 100 % off (round 6, scripts/dead_debug.py: 8e-3 on the features).  Those weights stay
            # as they are: what the matrix pipe multiplies is then fp16(s w), 3e-8 off at most.
            keep = (s == 0) | (np.abs(folded) < 2.0 ** -14)
            out[k] = np.where(keep, v, r / np.where(s != 0, s, 1)).astype(np.float32)
        else:
            out[k] = r
        if bias_correction and input_means is not None and k in input_means and np.asarray(input_means[k]).ndim == 2:
            # the mean input of the convolution in the model's own units: relu(bn(x)) = sw clamp(x) + tc in front of a dense
            # layer's 1x1, the normalised pixel in front of the stem
            abar = np.asarray(input_means[k], np.float64).mean(0)
            if bn and bn + "_gamma" in params and f"{m.group(1)}batchnorm{int(m.group(2))}_gamma" in params:
                _, _, sw1, tc1 = bn_relu_clamp_fold(params, f"{m.group(1)}batchnorm{int(m.group(2))}")
                abar = sw1.astype(np.float64) * abar + tc1
            elif k.endswith("conv0_weight") and v.shape[1:] == (3, 7, 7):
                abar = abar / (255.0 * np.array([0.229, 0.224, 0.225]))
            bias[k] = (out[k].astype(np.float64) - v.astype(np.float64)).sum((2, 3)) @ abar      # (taps see the same mean: borders ignored)
    if bias and prefix is not None:
        _apply_bias_correction(out, bias, prefix)
    return out


def make_densenet121_weights(seed: int = 0, prefix: str = "densenet0_", in_channels: int = 3,
                             fp16_model: bool = True):
    """Seeded DenseNet-121 ``.features`` parameters (6.87 M conv weights + 121 BN).
    With ``fp16_model`` the conv weights are fp16-representable (see as_fp16_model)."""
    rng = np.random.default_rng(seed)
    convs, final_bn, cfin = densenet121_layout()
    p = {}
    for cv in convs:
        k = {"stem": 7, "dense1x1": 1, "dense3x3": 3, "trans": 1}[cv["kind"]]
        cin = in_channels if cv["kind"] == "stem" else cv["cin"]
        fan_in = cin * k * k
        w = rng.normal(0.0, np.sqrt(2.0 / fan_in), (cv["cout"], cin, k, k)).astype(np.float32)
        p[prefix + cv["name"] + "_weight"] = w
        if cv["kind"] == "stem":
            # BN follows the stem conv; normalised uniform-u8 pixels have E[x^2]~1.7
            p.update(_bn(rng, prefix + cv["bn"], cv["cout"], var_center=3.4))
        else:
            p.update(_bn(rng, prefix + cv["bn"], cin))
    p.update(_bn(rng, prefix + final_bn, cfin))
    return as_fp16_model(p) if fp16_model else p


def make_dense_weights(seed: int, units: int, in_units: int, prefix: str):
    """``nn.Dense`` parameters (reference definitions.py:25,58,60,101)."""
    rng = np.random.default_rng(seed)
    return {
        prefix + "weight": rng.uniform(-0.07, 0.07, (units, in_units)).astype(np.float32),
        prefix + "bias": rng.uniform(-0.1, 0.1, units).astype(np.float32),
    }


def make_rnn_weights(seed: int, mode: str, input_size: int, hidden: int, prefix: str,
                     bidirectional: bool = True):
    """``mx.gluon.rnn.GRU/LSTM`` single-layer parameters (reference definitions.py:94-96).

    Gate order [EXT, SURVEY App. B]: GRU ``[r, z, n]``, LSTM ``[i, f, g, o]``.
    """
    g = {"gru": 3, "lstm": 4}[mode]
    rng = np.random.default_rng(seed)
    p = {}
    for d in (["l", "r"] if bidirectional else ["l"]):
        si = 1.0 / np.sqrt(input_size)
        sh = 1.0 / np.sqrt(hidden)
        p[f"{prefix}{d}0_i2h_weight"] = rng.uniform(-si, si, (g * hidden, input_size)).astype(np.float32)
        p[f"{prefix}{d}0_h2h_weight"] = rng.uniform(-sh, sh, (g * hidden, hidden)).astype(np.float32)
        p[f"{prefix}{d}0_i2h_bias"] = rng.uniform(-sh, sh, g * hidden).astype(np.float32)
        p[f"{prefix}{d}0_h2h_bias"] = rng.uniform(-sh, sh, g * hidden).astype(np.float32)
    return p


def make_gnmt_weights(seed: int, cell_type: str, input_size: int, hidden: int, embed: int, vocab: int,
                      num_layers: int = 2, num_bi_layers: int = 1, prefix: str = "gnmt_"):
    """Parameters of the reference captioner (gnmt.py:71-111,197-222; train_gnmt.py:211-233).

    Encoder: ``num_bi_layers`` bidirectional cell layers then uni-directional
    cell layers; decoder: ``num_layers`` cells, first takes ``[embed, H]``,
    others ``[H, H]``; scaled-Luong attention has one bias-free key projection
    ``(H, H)`` [EXT]; ``tgt_proj`` Dense(V) with bias; ``tgt_embed`` (V, embed).
    Default init in the reference is ``Uniform(0.1)`` (train_gnmt.py:231) with
    ``LSTMBias(1.0)`` for i2h biases (gnmt.py:410).
    """
    g = {"gru": 3, "lstm": 4}[cell_type]
    rng = np.random.default_rng(seed)
    u = lambda *s: rng.uniform(-0.1, 0.1, s).astype(np.float32)
    p = {}

    def cell(pref, fin):
        p[pref + "i2h_weight"] = u(g * hidden, fin)
        p[pref + "h2h_weight"] = u(g * hidden, hidden)
        p[pref + "i2h_bias"] = u(g * hidden)
        p[pref + "h2h_bias"] = u(g * hidden)

    fin = input_size
    for i in range(num_layers):
        if i < num_bi_layers:
            cell(f"{prefix}enc_rnn{i}_l_", fin)
            cell(f"{prefix}enc_rnn{i}_r_", fin)
            fin = 2 * hidden
        else:
            cell(f"{prefix}enc_rnn{i}_", fin)
            fin = hidden
    for i in range(num_layers):
        cell(f"{prefix}dec_rnn{i}_", (embed + hidden) if i == 0 else 2 * hidden)
    p[prefix + "dec_attention_key_weight"] = u(hidden, hidden)
    p[prefix + "tgt_proj_weight"] = u(vocab, hidden)
    p[prefix + "tgt_proj_bias"] = u(vocab)
    emb = rng.normal(0.0, 1.0, (vocab, embed)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)  # rows L2-normalised like data/embeddings-ex.txt
    emb[:4] = 0.0  # <unk>,<pad>,<bos>,<eos> get zero vectors (Vocab.set_embedding)
    p[prefix + "tgt_embed_weight"] = emb
    return p


def synthetic_frames_u8(n: int, size: int = 224, seed: int = 1234) -> np.ndarray:
    """NHWC uint8 frames, uniform [0,255] (SURVEY §8d synthetic inputs)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, size, size, 3), dtype=np.uint8)


IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
# csrc/common.h::stem_wfactor: what multiplies conv0's weights of input channel c before they are rounded to fp16
STEM_WFACTOR = np.array([64.0 / (255.0 * 0.229), 64.0 / (255.0 * 0.224), 64.0 / (255.0 * 0.225)]).astype(np.float32)


def normalize_to_nchw_f32(frames_u8: np.ndarray) -> np.ndarray:
    """``ToTensor`` + ``Normalize`` of the reference test transform (evaluate.py:96-97)."""
    x = frames_u8.astype(np.float32) / 255.0
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))
