"""CPU attribution study of the fp16 ACTIVATION path: where is the coherent error of flat frames born?

The DenseNet-121 .features graph in fp32 torch with the kernels' rounding points injected one group at a time:

    in    normalised input -> fp16 (stem MFMA operand)
    stem  stem + BN + ReLU + maxpool -> stored fp16
    bn1   relu(BN1(x)) -> fp16 (1x1 MFMA operand), per dense layer
    bott  relu(1x1 + shift) -> fp16 (3x3 MFMA operand), per dense layer
    new   3x3 output -> stored fp16 (the 32 new channels), per dense layer
    tin   transition: avg2x2(relu(BN(x))) -> fp16 (1x1 MFMA operand)
    tout  transition output -> stored fp16

Each point can be rounded to nearest ("rn") or with a position-keyed dither ("sr": stochastic rounding whose random number is a
hash of (y, x, channel, site) - the same for every frame, so a frame's features do not depend on its neighbours in the batch).
The oracle evaluates the SAME weights in fp32 without any rounding; feature and logit errors per family are printed.

    python scripts/round_study.py --exp attribute | dither | modes
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tennis_amd import weights as W          # noqa: E402
from tennis_amd import calib_frames as CF    # noqa: E402

POINTS = ["in", "stem", "bn1", "bott", "new", "tin", "tout"]
FLAT = ["constant", "halfblack", "saturated", "text", "stripes", "bright"]


def _hash_u13(shape, site):
    """(C, H, W) -> 13-bit pseudo-random integers from (channel, y, x, site); cheap multiplicative hash (what a kernel would do)"""
    C, H, Wd = shape
    c = torch.arange(C, dtype=torch.int64).view(C, 1, 1)
    y = torch.arange(H, dtype=torch.int64).view(1, H, 1)
    x = torch.arange(Wd, dtype=torch.int64).view(1, 1, Wd)
    h = (x * 0x9E3779B1 + y * 0x85EBCA77 + c * 0xC2B2AE3D + site * 0x27D4EB2F) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 12
    h = (h * 0x297A2D39) & 0xFFFFFFFF
    h ^= h >> 15
    return (h & 0x1FFF).to(torch.int32)


def round_fp16(x, mode, site=0, per_channel=True):
    if mode == "none":
        return x
    if mode == "rn":
        return x.half().float()
    if mode == "sr":      # add 13 random bits below the fp16 mantissa, truncate (exact for normal fp16 range)
        r = _hash_u13(x.shape[1:] if per_channel else (1,) + tuple(x.shape[2:]), site)
        b = x.contiguous().view(torch.int32)
        b = (b + r.unsqueeze(0)) & ~0x1FFF
        return b.view(torch.float32)
    raise ValueError(mode)


class Net:
    def __init__(self, p, pre="densenet0_"):
        self.p = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in p.items() if k.startswith(pre)}
        self.pre = pre

    def _bn(self, x, n):
        p = self.p
        return F.batch_norm(x, p[n + "_running_mean"], p[n + "_running_var"], p[n + "_gamma"], p[n + "_beta"], False, 0.0, 1e-5)

    @torch.no_grad()
    def __call__(self, x, modes=None, stages=(0, 1, 2, 3, 4), per_channel=True):
        """modes: {point: "rn"|"sr"}; stages: where the per-layer points are active (0 = stem, 1..4 = blocks / the transition behind)"""
        p, pre = self.p, self.pre
        modes = modes or {}
        site = [0]

        def rnd(v, point, st):
            site[0] += 1
            if point in modes and st in stages:
                return round_fp16(v, modes[point], site[0], per_channel)
            return v
        x = rnd(x, "in", 0)
        x = F.conv2d(x, p[pre + "conv0_weight"], stride=2, padding=3)
        x = F.max_pool2d(F.relu(self._bn(x, pre + "batchnorm0")), 3, 2, 1)
        x = rnd(x, "stem", 0)
        outer = 1
        for st, nl in enumerate((6, 12, 24, 16), 1):
            sp = f"{pre}stage{st}_"
            for li in range(nl):
                a = rnd(F.relu(self._bn(x, f"{sp}batchnorm{2 * li}")), "bn1", st)
                y = F.conv2d(a, p[f"{sp}conv{2 * li}_weight"])
                b = rnd(F.relu(self._bn(y, f"{sp}batchnorm{2 * li + 1}")), "bott", st)
                y = F.conv2d(b, p[f"{sp}conv{2 * li + 1}_weight"], padding=1)
                x = torch.cat([x, rnd(y, "new", st)], 1)
            if st != 4:
                a = F.avg_pool2d(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), 2, 2)      # (the kernels pool first: linear, same thing)
                a = rnd(a, "tin", st)
                x = rnd(F.conv2d(a, p[f"{pre}conv{outer}_weight"]), "tout", st)
                outer += 1
        x = F.avg_pool2d(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), 7)
        return x.flatten(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", default="attribute")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--families", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    p = W.make_densenet121_weights(0, fp16_model=True)
    net = Net(p)
    rng = np.random.default_rng(5)
    wd = torch.from_numpy(rng.normal(0, 1.0 / 32, (11, 1024)).astype(np.float32))          # a Dense(11) head of the usual scale
    fams = a.families.split(",") if a.families else CF.FAMILIES + CF.HELD_OUT
    xs = {f: torch.from_numpy(W.normalize_to_nchw_f32(CF.frames(f, a.frames, 224, seed=99))) for f in fams}
    ref = {f: net(xs[f]) for f in fams}
    for f in fams:
        print("ref %-12s |feat| max %.2f mean %.3f" % (f, ref[f].abs().max(), ref[f].abs().mean()))
    res = {}

    def run(tag, modes, **kw):
        t0 = time.time()
        row = {}
        for f in fams:
            got = net(xs[f], modes, **kw)
            row[f] = (float((got - ref[f]).abs().max()), float(((got - ref[f]) @ wd.T).abs().max()))
        res[tag] = row
        print("%-34s " % tag + " ".join("%s %.1e/%.0e" % (f[:4], *row[f]) for f in fams) +
              "  worst %.2e (%.0fs)" % (max(v[0] for v in row.values()), time.time() - t0), flush=True)

    ALL = {k: "rn" for k in POINTS}
    if a.exp == "attribute":
        run("all points rn", ALL)
        for pt in POINTS:
            run(f"only {pt}", {pt: "rn"})
        for st in (1, 2, 3, 4):
            run(f"all points, stage {st} only", ALL, stages=(st,))
        run("all points, stem only", ALL, stages=(0,))
    elif a.exp == "dither":
        run("all rn", ALL)
        run("stored sr (stem,new,tout)", dict(ALL, stem="sr", new="sr", tout="sr"))
        run("stored sr, pixel-only key", dict(ALL, stem="sr", new="sr", tout="sr"), per_channel=False)
        run("stored+bott sr", dict(ALL, stem="sr", new="sr", tout="sr", bott="sr"))
        run("stored+bott+tin sr", dict(ALL, stem="sr", new="sr", tout="sr", bott="sr", tin="sr"))
        run("everything sr", {k: "sr" for k in POINTS})
        run("no bn1 rounding, rest sr", {k: "sr" for k in POINTS if k != "bn1"})
        run("no bn1 rounding, rest rn", {k: "rn" for k in POINTS if k != "bn1"})
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
