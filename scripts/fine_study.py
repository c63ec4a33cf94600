"""CPU study (round 6): where is the calibrated conversion's error born on FINE CHECKERBOARDS (cells of 1 - 3 px, full contrast:
the frames with the 1.2e-3 tail of profiles/r05_parity_wide.json)?  fp32 torch graph, no kernel noise: for every layer group g
the model with g's convolutions converted and everything else exact is compared with the all-exact model.

    python scripts/fine_study.py [--frames 12] [--builtin 144] [--extra-fine N]   (N fine checkerboards added to the calibration set)
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
from conv_study import operand_means         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--builtin", type=int, default=144)
    ap.add_argument("--extra-fine", type=int, default=0)
    ap.add_argument("--wseed", type=int, default=0)
    a = ap.parse_args()
    torch.set_num_threads(8)
    p = W.make_densenet121_weights(a.wseed, fp16_model=False)
    ref_net = CS.Net(p)
    ev = {"fine": CF.fine_checkerboards(a.frames, 224, seed=5), "checker": CF.frames("checker", 4, 224, seed=2024), "photo": CF.frames("photo", 2, 224, seed=99),
          "constant": CF.frames("constant", 2, 224, seed=99)}
    ev = {k: torch.from_numpy(W.normalize_to_nchw_f32(v)) for k, v in ev.items()}
    ref = {f: ref_net(x).numpy() for f, x in ev.items()}
    cal = CF.default_calibration_frames(224, a.builtin)
    if a.extra_fine:
        cal = np.concatenate([cal, CF.fine_checkerboards(a.extra_fine, 224, seed=777)])
    fm = operand_means(ref_net, p, cal)
    q = W.as_fp16_model(p, input_means=fm)
    groups = {"stem": lambda k: k.endswith("densenet0_conv0_weight")}
    for t in (1, 2, 3):
        groups[f"trans{t}"] = (lambda t: lambda k: k.endswith(f"densenet0_conv{t}_weight"))(t)
    for b in (1, 2, 3, 4):
        groups[f"b{b} 1x1"] = (lambda b: lambda k: re.search(rf"stage{b}_conv\d+_weight", k) and int(re.search(r"conv(\d+)_", k).group(1)) % 2 == 0)(b)
        groups[f"b{b} 3x3"] = (lambda b: lambda k: re.search(rf"stage{b}_conv\d+_weight", k) and int(re.search(r"conv(\d+)_", k).group(1)) % 2 == 1)(b)
    groups["all"] = lambda k: True
    groups["all but stem"] = lambda k: not k.endswith("densenet0_conv0_weight")
    groups["all but stem, b1"] = lambda k: not (k.endswith("densenet0_conv0_weight") or "stage1_" in k)
    groups["all but stem, b1 1x1"] = lambda k: not (k.endswith("densenet0_conv0_weight") or (re.search(r"stage1_conv\d+_weight", k) and int(re.search(r"conv(\d+)_", k).group(1)) % 2 == 0))

    def row(tag, model):
        net = CS.Net(model)
        e = {f: net(x).numpy() - ref[f] for f, x in ev.items()}
        print("%-22s " % tag + " ".join("%s %.2e (>1e-3: %d)" % (f[:5], np.abs(e[f]).max(), (np.abs(e[f]) > 1e-3).sum()) for f in ev), flush=True)

    # a partial conversion: the group's convolutions calibrated (and ONLY their bias correction in the BatchNorm running means), every
    # other convolution exact
    isconv = lambda k: k.endswith("_weight") and p[k].ndim == 4
    for name, sel in groups.items():
        qg = q if name == "all" else W.as_fp16_model(p, input_means={k: v for k, v in fm.items() if sel(k)})
        row(name, {k: (p[k] if (isconv(k) and not sel(k)) else v) for k, v in qg.items()})


if __name__ == "__main__":
    main()
