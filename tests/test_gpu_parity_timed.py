"""-m gpu: parity of the configuration bench.py times, measured on the kernels bench.py times.

One 256-frame call through a `max_batch=256` encoder = the strip kernels (56x56, 28x28 K <= 320), the streamed 14x14 block, the
LDS-resident 7x7 block, the two-stream split - NOT the small-batch tile kernels the 16-family matrix of
tests/test_gpu_calibration.py runs.  Input: 15 frames of each of the 16 families + 16 fine checkerboards (cells of 1 - 3 px, full
contrast: where round 5's wide evaluation found its 11 values over the bar).  Oracle: oracle/torch_ref.py on the UN-rounded fp32
weights and the un-rounded normalised input (reference models/vision/definitions.py:27-33).  Two parameter sets: bench.py's seeded
fp32 weights, and a "trained-looking" set (tests/tools/trained_like.py: correlated filters, log-normal BatchNorm gammas with
near-dead and negative channels, measured running statistics).  Two modes: the timed one (calibrated conversion to one fp16
number per weight) and the exact-weights mode.  Everything measured goes to gpurun_out/parity_timed.json."""
import json
import os

import numpy as np
import pytest
import torch

from tools import parity_timed as PT

pytestmark = pytest.mark.gpu
BAR = 1e-3

_results = {}


def _dump():
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"oracle": "oracle/torch_ref.py: fp32 graph, un-rounded fp32 weights, un-rounded normalised input", "bar": BAR,
               "batch": "256 frames in ONE call of a max_batch=256 encoder (default kernels: strip / streamed / LDS-resident, two-stream split)",
               "runs": _results}, open("gpurun_out/parity_timed.json", "w"), indent=1)


@pytest.fixture(scope="module")
def refs():
    return {}


@pytest.mark.parametrize("kind", ["seeded", "trained"])
@pytest.mark.parametrize("mode", ["calibrated", "exact"])
def test_timed_path_parity(kind, mode, refs, report):
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    r, ref = PT.measure(kind, mode, 256, ref=refs.get(kind))
    refs[kind] = ref
    _results[f"{kind} / {mode}"] = r
    _dump()
    tag = f"timed_path_{kind}_{mode}"
    report[tag + "_feature_max"] = r["feature_max"]
    report[tag + "_logit_max"] = r["logit_max"]
    report[tag + "_values_over_bar"] = r["over_bar"]
    report[tag + "_feature_max_scaled"] = r["feature_max_scaled"]
    print(kind, mode, {k: r[k] for k in ("feature_max", "logit_max", "over_bar", "feature_max_scaled", "over_bar_scaled", "worst_family")})
    if mode == "calibrated":
        assert r["timed_kernels_ran"], r["kernel_families"]         # the strip / streamed / LDS-resident families, not the tile kernels
    fams = r["families"]
    sixteen = [f for f in fams if f != "finechecker"]
    if kind == "seeded":
        # north_star's bar - "logits within 1e-3" - on every frame of the 16 families, in both modes
        assert max(fams[f]["logit_max"] for f in sixteen) < BAR, {f: fams[f]["logit_max"] for f in sixteen}
        over = sum(fams[f]["over_bar"] for f in sixteen)
        worst = max(fams[f]["feature_max"] for f in sixteen)
        report[tag + "_16_families_feature_max"] = worst
        report[tag + "_16_families_values_over_bar"] = over
        if mode == "exact":
            assert worst < BAR and over == 0, (worst, over)          # measured 6.8e-4
        else:
            # the timed configuration: 245 760 feature values, measured 2 over the bar (1.07e-3 checker, 1.16e-3 text) - a tail of the
            # weight conversion on piecewise-flat frames; pinned here, stated in DESIGN.md section 4 and in bench.py's parity_live
            assert over <= 8 and worst < 1.5e-3, (worst, over)
        # fine checkerboards (cells of 1 - 3 px, full contrast) are beyond the bar in EVERY mode (exact weights: 1.5e-3): the fp16
        # activation path itself; the bound pins the tail
        assert fams["finechecker"]["feature_max"] < (2.5e-3 if mode == "exact" else 6e-3), fams["finechecker"]
    else:
        # Trained-looking parameters (heavy-tailed BatchNorm scales, near-dead and negative gammas, measured running statistics)
        # amplify every rounding of the fp16 activation path 3 - 10 x more than the seeded ones; no mode holds 1e-3 there
        # (DESIGN.md section 4 "trained-looking weights").  What the test pins: natural content stays within a few 1e-3, nothing
        # is catastrophic (round 6 found and fixed two 1e-2 .. 5e-1 failures of exactly this kind: the constant of a near-dead
        # stem channel rounded in fp16, and the clamp form's constant cancelling against a weight in fp16's subnormals)
        natural = ["noise", "blobs", "scene", "photo", "jpeg"]
        assert max(fams[f]["feature_max_scaled"] for f in natural) < 6e-3, {f: fams[f]["feature_max_scaled"] for f in natural}
        assert fams["noise"]["feature_max_scaled"] < 4e-3
        assert r["feature_max_scaled"] < 0.15 and np.isfinite(r["feature_max"]), (r["worst_family"], r["feature_max_scaled"])


def test_near_dead_batchnorm_channels(report):
    """7 % of the gammas of one BatchNorm group scaled by 1e-2 .. 1e-6 (what trained checkpoints contain), seeded weights otherwise:
    the exact-weights mode stays at the seeded model's error whatever the sign of the dead channels' beta (round 6: 4.8e-3 with
    negative betas before the clamp form's constant was split at clamp(0, lo, hi): csrc/api.hip), and so do the default kernels
    on the plainly converted model."""
    import re
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    base = W.make_densenet121_weights(0, fp16_model=False)
    frames = W.synthetic_frames_u8(4, 224, seed=9)
    x, xd = torch.from_numpy(W.normalize_to_nchw_f32(frames)), torch.from_numpy(frames).cuda()
    groups = {"bn0": lambda k: k == "densenet0_batchnorm0_gamma", "bn1": lambda k: bool(re.search(r"stage\d+_batchnorm\d*[02468]_gamma", k)),
              "bn2": lambda k: bool(re.search(r"stage\d+_batchnorm\d*[13579]_gamma", k))}
    for name, sel in groups.items():
        for sign in ((-1.0, 1.0) if name == "bn1" else (0.0,)):
            rng = np.random.default_rng(3)
            p = dict(base)
            for k in sorted(p):
                if k.endswith("_gamma") and sel(k):
                    g = p[k].copy()
                    dead = rng.random(g.size) < 0.07
                    g[dead] *= 10.0 ** rng.uniform(-6, -2, int(dead.sum()))
                    p[k] = g.astype(np.float32)
                    if sign:
                        bk = k.replace("_gamma", "_beta")
                        b = p[bk].copy()
                        b[dead] = sign * np.abs(b[dead])
                        p[bk] = b
            ref = TorchDenseNet121(p)(x).numpy()
            q = W.as_fp16_model(p)
            refq = TorchDenseNet121(q)(x).numpy()
            e_exact = float(np.abs(DenseNet121Features(p, 224, max_batch=4, exact_weights=True)(xd).cpu().numpy() - ref).max())
            e_plain = float(np.abs(DenseNet121Features(q, 224, max_batch=4)(xd).cpu().numpy() - refq).max())
            report[f"dead_{name}_{sign:+.0f}_exact_mode_err"] = e_exact
            report[f"dead_{name}_{sign:+.0f}_kernels_alone_err"] = e_plain
            assert e_exact < BAR and e_plain < BAR, (name, sign, e_exact, e_plain)


def test_a_frame_does_not_depend_on_its_batch():
    """the 256-frame call and a 128-frame call (two half batches of 64: still the strip / streamed kernels - 64 frames per launch is
    where they switch in) give the same bits for the same frame, and so does the pipelined whole-batch form (round 6)"""
    from tennis_amd import weights as W
    p = W.make_densenet121_weights(0)
    frames, _ = PT.batch(256)
    x = torch.from_numpy(frames).cuda()
    from tennis_amd.engine import DenseNet121Features
    enc = DenseNet121Features(p, 224, max_batch=256)
    big = enc(x)
    small = DenseNet121Features(p, 224, max_batch=128)(x[128:256].contiguous())
    assert torch.equal(big[128:256], small)
    enc.set_pipelined(True)
    outs = [torch.empty_like(big) for _ in range(3)]
    for o in outs:
        enc(x, out=o)
    enc.join(0); enc.join(1)
    enc.set_pipelined(False)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, big)


def test_family_parity_at_the_reference_default_input_size(report):
    """data_shape defaults to 512 in the reference's drivers (train.py:44, evaluate.py:42; 4096-d features, train.py:259).  The same
    family matrix there: 32 frames (two of each of the 15 generated families, fine checkerboards) in ONE call of a max_batch=32
    encoder - the strip kernels on the 128x128 / 64x64 maps (from 13 / 22 frames on), the tile kernels behind them - with the
    calibrated conversion of the seeded fp32 weights, against the fp32 oracle on the un-rounded weights and input."""
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    r, ref512 = PT.measure("seeded", "calibrated", 32, size=512)
    _results["seeded / calibrated / 512x512 / batch 32"] = r
    _dump()
    fams = r["families"]
    sixteen = [f for f in fams if f != "finechecker"]
    report["parity_512_feature_max_16_families"] = max(fams[f]["feature_max"] for f in sixteen)
    report["parity_512_logit_max_16_families"] = max(fams[f]["logit_max"] for f in sixteen)
    print({f: (fams[f]["feature_max"], fams[f]["logit_max"]) for f in fams})
    assert "dense_layer_strip_128x128" in r["kernel_families"] and "dense_layer_strip_64x64" in r["kernel_families"], r["kernel_families"]
    assert r["values"] == 32 * 4096
    # Measured (round 6): textured / natural families 4.5e-4 .. 8.8e-4, piecewise-flat ones up to 1.9e-3 (`constant`: 81 of 8 192 values
    # over 1e-3; `text` 1.15e-3; `halfblack` 1.0e-3) - at 5.2 x the pixels per map a flat region makes its coherent rounding error in
    # 5.2 x the places, and the 4096-wide Dense(11) sums four times the feature errors.  512 x 512 is NOT inside the bar on flat
    # frames; the test holds the natural families to it and pins the rest.
    natural = ["noise", "lowcontrast", "gradient", "blobs", "scene", "stripes", "dark", "bright", "tinted", "saturated", "photo", "jpeg"]
    assert max(fams[f]["feature_max"] for f in natural) < BAR, {f: fams[f]["feature_max"] for f in natural}
    assert max(fams[f]["logit_max"] for f in natural) < BAR, {f: fams[f]["logit_max"] for f in natural}
    assert max(fams[f]["feature_max"] for f in sixteen) < 2.5e-3 and max(fams[f]["logit_max"] for f in sixteen) < 2.5e-3

    # Round 6: the exact-weights mode at 512 x 512 (VERDICT r5 "missing" 3: it did not exist there) - the 128 x 128 block on the
    # un-fused layer kernels with the hi + lo pass (conv1x1.hip / conv3x3.hip EX), the 64 / 32 / 16 maps on the tile kernel's.
    rx, _ = PT.measure("seeded", "exact", 32, ref=ref512, size=512)
    _results["seeded / exact / 512x512 / batch 32"] = rx
    _dump()
    fx = rx["families"]
    report["parity_512_exact_feature_max_16_families"] = max(fx[f]["feature_max"] for f in sixteen)
    report["parity_512_exact_logit_max_16_families"] = max(fx[f]["logit_max"] for f in sixteen)
    print("exact", {f: (fx[f]["feature_max"], fx[f]["logit_max"]) for f in fx})
    assert "conv1x1_bnrelu" in rx["kernel_families"] and "dense_layer_fused_64x64" in rx["kernel_families"], rx["kernel_families"]
    # Measured: 14 families <= 9.3e-4, `constant` 1.15e-3 (7 of 8 192 values over), `checker` 1.02e-3 (one value); logits <= 1.04e-3
    assert max(fx[f]["feature_max"] for f in natural) < BAR, {f: fx[f]["feature_max"] for f in natural}
    assert max(fx[f]["feature_max"] for f in sixteen) < 1.5e-3 and max(fx[f]["logit_max"] for f in sixteen) < 1.5e-3
    assert sum(fx[f]["over_bar"] for f in sixteen) <= 24


@pytest.mark.parametrize("size", [236, 448])
def test_exact_weights_mode_at_other_input_sizes(size, report):
    """Round 6: TN_ENC_EXACT_WEIGHTS is no longer tied to the 224 x 224 maps.  236: 59 / 29 / 14 / 7 maps (the first two blocks on
    the un-fused layer kernels with the hi + lo pass); 448: 112 un-fused, 56 x 56 with K up to 480 (tile kernel to K = 256,
    un-fused past it), 28 / 14 on the tile kernel.  Four frames (noise, lowcontrast, constant, gradient) against the fp32 oracle on
    the un-rounded weights.  Measured: 236 - all four <= 5.8e-4; 448 - the three textured ones <= 4.6e-4, `constant` 1.25e-3 (3 of
    4 096 values over 1e-3: the flat-frame tail of the fp16 ACTIVATION path, which grows with the number of pixels per map - 1.15e-3
    at 512 x 512 in this mode, 1.9e-3 with one fp16 number per weight)."""
    r, _ = PT.measure("seeded", "exact", 4, size=size)
    _results[f"seeded / exact / {size}x{size} / batch 4"] = r
    _dump()
    report[f"parity_{size}_exact_feature_max"] = r["feature_max"]
    print(size, r["feature_max"], r["logit_max"], r["kernel_families"])
    assert "conv1x1_bnrelu" in r["kernel_families"] and "conv3x3_bnrelu" in r["kernel_families"], r["kernel_families"]
    fams = r["families"]
    textured = [f for f in fams if f != "constant"]
    assert max(fams[f]["feature_max"] for f in textured) < BAR and max(fams[f]["logit_max"] for f in textured) < BAR, fams
    assert fams["constant"]["feature_max"] < (BAR if size < 400 else 1.6e-3) and fams["constant"]["over_bar"] <= 8, fams["constant"]
