"""CPU study: which layer groups carry the error of the CALIBRATED fp16 conversion (round 5)?  fp32 torch graph, no kernel noise:
for every group g the model with g's convolutions converted and everything else exact is compared with the all-exact model.

    python scripts/conv_study.py [--frames 2] [--builtin 72]
"""
import argparse
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tennis_amd import weights as W          # noqa: E402
from tennis_amd import calib_frames as CF    # noqa: E402
import calib_study as CS                     # noqa: E402


def operand_means(net, p, frames_u8):
    """what tennis_amd.calibrate.frame_means returns, from the CPU graph: the per-frame mean of every convolution's OPERAND -
    clamp(x) for the dense layers' 1x1 (csrc/calib_host.hip), x - 255 mean for the stem, relu(bn(x)) elsewhere"""
    fm = CS.frame_means(net, frames_u8)
    out = {}
    for k, v in fm.items():
        m = re.fullmatch(r"(.*stage\d+_)conv(\d+)_weight", k)
        if m and int(m.group(2)) % 2 == 0:
            lo, hi, sw, tc = W.bn_relu_clamp_fold(p, f"{m.group(1)}batchnorm{m.group(2)}")
            out[k] = np.where(sw != 0, (v - tc) / np.where(sw != 0, sw, 1), 0.0)
        elif k.endswith("densenet0_conv0_weight"):
            out[k] = v * (255.0 * np.array([0.229, 0.224, 0.225]))
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--builtin", type=int, default=72)
    a = ap.parse_args()
    torch.set_num_threads(8)
    p = W.make_densenet121_weights(0, fp16_model=False)
    ref_net = CS.Net(p)
    evf = CF.FAMILIES + CF.HELD_OUT
    ev = {f: torch.from_numpy(W.normalize_to_nchw_f32(CF.frames(f, a.frames, 224, seed=99))) for f in evf}
    ref = {f: ref_net(ev[f]).numpy() for f in evf}
    fm = operand_means(ref_net, p, CF.default_calibration_frames(224, a.builtin))
    q = W.as_fp16_model(p, input_means=fm)
    plain = W.as_fp16_model(p)
    groups = {"stem": lambda k: k.endswith("densenet0_conv0_weight")}
    for t in (1, 2, 3):
        groups[f"trans{t}"] = (lambda t: lambda k: k.endswith(f"densenet0_conv{t}_weight"))(t)
    for b in (1, 2, 3, 4):
        groups[f"b{b} 1x1"] = (lambda b: lambda k: re.search(rf"stage{b}_conv\d+_weight", k) and int(re.search(r"conv(\d+)_", k).group(1)) % 2 == 0)(b)
        groups[f"b{b} 3x3"] = (lambda b: lambda k: re.search(rf"stage{b}_conv\d+_weight", k) and int(re.search(r"conv(\d+)_", k).group(1)) % 2 == 1)(b)
    groups["all"] = lambda k: True

    def row(tag, model):
        net = CS.Net(model)
        e = {f: net(ev[f]).numpy() - ref[f] for f in evf}
        print("%-10s " % tag + " ".join("%s %.1e" % (f[:4], np.abs(e[f]).max()) for f in evf) +
              "  worst %.2e  rms %.2e" % (max(np.abs(v).max() for v in e.values()), np.sqrt(np.mean([np.mean(v ** 2) for v in e.values()]))), flush=True)

    for name, sel in groups.items():
        row(name, {k: (q[k] if (k.endswith("_weight") and sel(k)) else v) for k, v in p.items()})
    row("all plain", plain)


if __name__ == "__main__":
    main()


def bias_correct(p, q, fm_unfolded, pre="densenet0_"):
    """q with the running means of every BatchNorm that consumes a convolution's output moved by that output's mean conversion
    error sum_k (q - p)[n, k] E[a_k] (a: the convolution's input, averaged over the calibration frames and their pixels)"""
    out = dict(q)
    cfg = (6, 12, 24, 16)
    cin = [64]
    for b in range(3):
        cin.append((cin[b] + 32 * cfg[b]) // 2)

    def bias(name):
        d = (q[name].astype(np.float64) - p[name].astype(np.float64)).sum((2, 3))          # (N, C): taps see the same mean (borders ignored)
        return d @ fm_unfolded[name].mean(0)

    def shift(bn, lo, b):
        k = pre + bn + "_running_mean"
        out[k] = out[k].copy()
        out[k][lo:lo + b.size] += b.astype(np.float32)

    shift("batchnorm0", 0, bias(pre + "conv0_weight"))
    for s in range(1, 5):
        nl = cfg[s - 1]
        term = f"batchnorm{s}"           # the BatchNorm behind the block: transition s, or the head (s = 4)
        if s > 1:                        # the transition in front of this block feeds channels [0, cin)
            b = bias(pre + f"conv{s - 1}_weight")
            for l in range(nl):
                shift(f"stage{s}_batchnorm{2 * l}", 0, b)
            shift(term, 0, b)
        for l in range(nl):
            shift(f"stage{s}_batchnorm{2 * l + 1}", 0, bias(pre + f"stage{s}_conv{2 * l}_weight"))
            b = bias(pre + f"stage{s}_conv{2 * l + 1}_weight")
            c0 = cin[s - 1] + 32 * l
            for l2 in range(l + 1, nl):
                shift(f"stage{s}_batchnorm{2 * l2}", c0, b)
            shift(term, c0, b)
    return out
