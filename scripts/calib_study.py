"""CPU study of the calibrated fp16 conversion OFF its calibration distribution.

Isolates the CONVERSION error: the fp32 torch graph (oracle/torch_ref.py restated with a hook) evaluates the converted
weights and the un-rounded weights on the same frames; no GPU, no kernel noise.  For every (calibration family, set size)
the pooled-feature error on every evaluation family is printed.

    python scripts/calib_study.py [--method mean|vec] [--rank R]
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


class Net:
    """DenseNet-121 .features in fp32 torch with an optional hook on every convolution input."""

    def __init__(self, p, pre="densenet0_"):
        self.p = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in p.items() if k.startswith(pre)}
        self.pre = pre

    def _bn(self, x, n):
        p = self.p
        return F.batch_norm(x, p[n + "_running_mean"], p[n + "_running_var"], p[n + "_gamma"], p[n + "_beta"], False, 0.0, 1e-5)

    @torch.no_grad()
    def __call__(self, x, hook=None):
        p, pre = self.p, self.pre

        def conv(a, name, **kw):
            if hook is not None:
                hook(pre + name + "_weight", a)
            return F.conv2d(a, p[pre + name + "_weight"], **kw)
        x = conv(x, "conv0", stride=2, padding=3)
        x = F.max_pool2d(F.relu(self._bn(x, pre + "batchnorm0")), 3, 2, 1)
        outer = 1
        for st, nl in enumerate((6, 12, 24, 16), 1):
            sp = f"stage{st}_"
            for li in range(nl):
                y = conv(F.relu(self._bn(x, f"{pre}{sp}batchnorm{2 * li}")), f"{sp}conv{2 * li}")
                y = conv(F.relu(self._bn(y, f"{pre}{sp}batchnorm{2 * li + 1}")), f"{sp}conv{2 * li + 1}", padding=1)
                x = torch.cat([x, y], 1)
            if st != 4:
                x = conv(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), f"conv{outer}")
                x = F.avg_pool2d(x, 2, 2)
                outer += 1
        x = F.avg_pool2d(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), 7)
        return x.flatten(1)


QUAD = False


def frame_means(net, frames_u8):
    """per-frame spatial mean of every convolution's input channels: {conv weight name: (n_frames, cin)}"""
    out = {}

    def hook(name, a):
        m = [a.mean((2, 3))]
        if QUAD:
            h, w = a.shape[2] // 2, a.shape[3] // 2
            m += [a[:, :, :h, :w].mean((2, 3)), a[:, :, :h, w:].mean((2, 3)), a[:, :, h:, :w].mean((2, 3)), a[:, :, h:, w:].mean((2, 3))]
        out[name] = torch.cat(m).double().numpy()
    x = torch.from_numpy(W.normalize_to_nchw_f32(frames_u8))
    net(x, hook)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="mean")
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--eval-frames", type=int, default=4)
    ap.add_argument("--sizes", default="8")
    ap.add_argument("--calib", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--quad", action="store_true")
    ap.add_argument("--ridge", type=float, default=0.05)
    ap.add_argument("--sweeps", type=int, default=3)
    ap.add_argument("--train-error", action="store_true", help="also evaluate on the calibration frames themselves")
    a = ap.parse_args()
    torch.set_num_threads(8)
    global QUAD
    QUAD = a.quad
    p = W.make_densenet121_weights(0, fp16_model=False)
    ref_net = Net(p)
    fams = CF.FAMILIES
    evf = CF.FAMILIES + CF.HELD_OUT
    ev = {f: CF.frames(f, a.eval_frames, 224, seed=99) for f in evf}
    ref = {f: ref_net(torch.from_numpy(W.normalize_to_nchw_f32(ev[f]))).numpy() for f in evf}
    for f in fams:
        print("ref", f, "feature absmax %.3f mean %.3f" % (np.abs(ref[f]).max(), np.abs(ref[f]).mean()))

    def evaluate(q, tag):
        net = Net(q)
        row = {}
        for f in evf:
            got = net(torch.from_numpy(W.normalize_to_nchw_f32(ev[f]))).numpy()
            row[f] = float(np.abs(got - ref[f]).max())
        print("%-28s " % tag + " ".join("%s %.1e" % (f[:4], row[f]) for f in evf) + "  worst %.2e" % max(row.values()), flush=True)
        return row

    res = {"plain": evaluate(W.as_fp16_model(p), "plain rounding")}
    calibs = a.calib.split(",") if a.calib else fams + ["mixed"]
    for cf in calibs:
        for n in [int(s) for s in a.sizes.split(",")]:
            t0 = time.time()
            if cf == "mixed":
                cal = CF.default_calibration_frames(224, n * len(fams))
            elif cf.startswith("loo:"):      # every family but one, n frames of each
                cal = np.concatenate([CF.frames(f, n, 224, seed=4321) for f in fams if f != cf[4:]])
            else:
                cal = CF.frames(cf, n, 224, seed=4321)
            fm = frame_means(ref_net, cal)
            if a.method == "mean":
                q = W.as_fp16_model(p, input_means={k: v.mean(0) for k, v in fm.items()})
            else:
                W._VEC_RIDGE, W._VEC_SWEEPS = a.ridge, a.sweeps
                q = W.as_fp16_model(p, input_means=fm)
                if a.train_error:
                    net = Net(q)
                    xc = torch.from_numpy(W.normalize_to_nchw_f32(cal))
                    print("   training error (the calibration frames themselves): %.2e" % float((net(xc) - ref_net(xc)).abs().max()))
            res[f"{cf}/{n}"] = evaluate(q, f"cal {cf}/{n} ({time.time() - t0:.0f}s)")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
