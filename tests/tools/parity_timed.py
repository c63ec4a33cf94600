"""(test infrastructure: imports oracle/)  Parity of the TIMED configuration where the throughput is measured.

Rounds 4-5 measured the 16-family parity matrix with `max_batch=2` / `8`: those calls run the small-batch TILE kernels.  The strip
kernels (56x56, 28x28 K <= 320), the streamed 14x14 block and the LDS-resident 7x7 block - what bench.py times - only switch in
from 64 frames per launch.  Here ONE call of 256 frames goes through the default kernels of a `max_batch=256` encoder (the test
asserts from the library's own launch statistics that the strip / streamed / LDS-resident families ran), the input is a mix of
every frame family + fine checkerboards + decoded JPEG fixtures, and the oracle is oracle/torch_ref.py on the UN-rounded fp32
weights and the un-rounded normalised input (reference models/vision/definitions.py:27-33).

  python tests/tools/parity_timed.py [--weights seeded,trained] [--modes calibrated,exact] [--out gpurun_out/parity_timed.json]
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

JPEG_NPZ = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
BAR = 1e-3
# the kernel families bench.py's batch runs on (tn_densenet121_profile names): a 256-frame call that did not use them is not the timed path
TIMED_FAMILIES = ("dense_layer_strip_56x56", "dense_layer_strip_28x28", "dense_block_stream_14x14", "dense_block_lds_7x7")


def make_weights(kind: str, seed: int = 0) -> dict:
    from tennis_amd import weights as W
    if kind == "seeded":
        return W.make_densenet121_weights(seed, fp16_model=False)          # bench.py's parameters: fp32, not fp16-representable
    if kind == "trained":
        from tools.trained_like import make_trained_like_weights
        return make_trained_like_weights(seed)
    raise ValueError(kind)


def batch(n: int = 256, seed: int = 5, size: int = 224):
    from tennis_amd import calib_frames as CF
    return CF.mixed_batch(n, size, seed, JPEG_NPZ)


def oracle_features(params32: dict, frames_u8: np.ndarray) -> np.ndarray:
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    net = TorchDenseNet121(params32)
    step = 64 if frames_u8.shape[1] <= 256 else 8
    out = [net(torch.from_numpy(W.normalize_to_nchw_f32(frames_u8[i:i + step]))).numpy() for i in range(0, len(frames_u8), step)]
    return np.concatenate(out)


def encoder_for(params32: dict, mode: str, max_batch: int = 256, ctx=None, size: int = 224):
    from tennis_amd.calibrate import calibrated_fp16_model
    from tennis_amd.engine import DenseNet121Features
    if mode == "exact":
        return DenseNet121Features(params32, size, max_batch=max_batch, exact_weights=True, ctx=ctx)
    if mode == "calibrated":
        return DenseNet121Features(calibrated_fp16_model(params32, None, size, ctx=ctx), size, max_batch=max_batch, ctx=ctx)
    raise ValueError(mode)


def summarize(feat: np.ndarray, ref: np.ndarray, labels, dense_w: np.ndarray) -> dict:
    """per family: max-abs error of features and Dense(11) logits, count over the bar; the same with the bar scaled for frames whose
    reference features leave the range the 1e-3 bar was set for (|f| <= 8: a feature of 140 cannot be carried to 7e-6 relative by
    ANY fp16-activation path, nor is the fp32 oracle itself reproducible to that between BLAS builds)"""
    e = feat.astype(np.float64) - ref
    el = e @ dense_w.astype(np.float64).T
    scale = np.maximum(1.0, np.abs(ref).max(1) / 8.0)
    fams = {}
    for f in dict.fromkeys(labels):
        idx = [i for i, l in enumerate(labels) if l == f]
        a = np.abs(e[idx])
        fams[f] = {"frames": len(idx), "feature_max": float(a.max()), "feature_rms": float(np.sqrt((a ** 2).mean())),
                   "logit_max": float(np.abs(el[idx]).max()), "over_bar": int((a > BAR).sum()),
                   "ref_abs_max": float(np.abs(ref[idx]).max()),
                   "feature_max_scaled": float((a / scale[idx, None]).max()), "over_bar_scaled": int((a / scale[idx, None] > BAR).sum()),
                   "logit_max_scaled": float((np.abs(el[idx]) / scale[idx, None]).max())}
    return {"values": int(e.size), "frames": int(e.shape[0]), "bar": BAR,
            "feature_max": float(np.abs(e).max()), "logit_max": float(np.abs(el).max()), "over_bar": int((np.abs(e) > BAR).sum()),
            "feature_max_scaled": max(v["feature_max_scaled"] for v in fams.values()),
            "logit_max_scaled": max(v["logit_max_scaled"] for v in fams.values()),
            "over_bar_scaled": sum(v["over_bar_scaled"] for v in fams.values()),
            "worst_family": max(fams, key=lambda f: fams[f]["feature_max_scaled"]), "families": fams}


def measure(kind: str, mode: str, n: int = 256, seed: int = 5, wseed: int = 0, ref=None, size: int = 224):
    from tennis_amd import weights as W
    p = make_weights(kind, wseed)
    frames, labels = batch(n, seed, size)
    if ref is None:
        ref = oracle_features(p, frames)
    enc = encoder_for(p, mode, max_batch=n, size=size)
    x = torch.from_numpy(frames).cuda()
    feat = enc(x).cpu().numpy()
    stats, _ = enc.profile(x)
    ran = sorted(s["name"] for s in stats)
    wd = W.make_dense_weights(1, 11, ref.shape[1], "framemodel0_dense0_")["framemodel0_dense0_weight"]
    out = summarize(feat, ref, labels, wd)
    out["kernel_families"] = ran
    out["timed_kernels_ran"] = all(f in ran for f in TIMED_FAMILIES) if (mode == "calibrated" and size == 224) else None
    del enc
    return out, ref


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="seeded,trained")
    ap.add_argument("--modes", default="calibrated,exact")
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--out", default="gpurun_out/parity_timed.json")
    a = ap.parse_args()
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    res = {"oracle": "oracle/torch_ref.py: fp32 graph, un-rounded fp32 weights, un-rounded normalised input", "bar": BAR, "frames": a.frames, "runs": {}}
    for kind in a.weights.split(","):
        ref = None
        for mode in a.modes.split(","):
            r, ref = measure(kind, mode, a.frames, ref=ref, size=a.size)
            res["runs"][f"{kind} / {mode}"] = r
            print(kind, mode, {k: r[k] for k in ("feature_max", "logit_max", "over_bar", "feature_max_scaled", "over_bar_scaled", "worst_family", "timed_kernels_ran")}, flush=True)
            for f, v in r["families"].items():
                print("   %-12s feat %.2e (scaled %.2e) logit %.2e over %d  |ref| %.1f" % (f, v["feature_max"], v["feature_max_scaled"], v["logit_max"], v["over_bar"], v["ref_abs_max"]), flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
