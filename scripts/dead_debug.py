"""(tuning tool) Which BatchNorm group's near-dead gammas break which mode?  Seeded weights, 7 % of one group's gammas scaled by 1e-2 .. 1e-6."""
import os, re, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
from oracle.torch_ref import TorchDenseNet121
torch.set_num_threads(32)
base = W.make_densenet121_weights(0, fp16_model=False)
frames = W.synthetic_frames_u8(4, 224, seed=9)
x = torch.from_numpy(W.normalize_to_nchw_f32(frames)); xd = torch.from_numpy(frames).cuda()
groups = {"none": lambda k: False, "bn0": lambda k: k == "densenet0_batchnorm0_gamma",
          "bn1": lambda k: bool(re.search(r"stage\d+_batchnorm\d*[02468]_gamma", k)), "bn2": lambda k: bool(re.search(r"stage\d+_batchnorm\d*[13579]_gamma", k)),
          "trans": lambda k: bool(re.fullmatch(r"densenet0_batchnorm[123]_gamma", k))}
for st in (1, 2, 3, 4):
    groups[f"bn1 stage{st}"] = (lambda st: lambda k: bool(re.search(rf"stage{st}_batchnorm\d*[02468]_gamma", k)))(st)
if len(sys.argv) > 1:
    groups = {k: v for k, v in groups.items() if k in sys.argv[1].split(",")}
SIGN = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # force the sign of the dead channels' beta
for name, sel in groups.items():
    rng = np.random.default_rng(3)
    p = dict(base)
    for k in sorted(p):
        if k.endswith("_gamma") and sel(k):
            g = p[k].copy(); dead = rng.random(g.size) < 0.07
            g[dead] *= 10.0 ** rng.uniform(-6, -2, int(dead.sum())); p[k] = g.astype(np.float32)
            if SIGN:
                bk = k.replace("_gamma", "_beta"); b = p[bk].copy(); b[dead] = SIGN * np.abs(b[dead]); p[bk] = b
    ref = TorchDenseNet121(p)(x).numpy()
    q = W.as_fp16_model(p)
    refq = TorchDenseNet121(q)(x).numpy()
    e_exact = np.abs(DenseNet121Features(p, 224, max_batch=4, exact_weights=True)(xd).cpu().numpy() - ref).max()
    e_plain = np.abs(DenseNet121Features(q, 224, max_batch=4)(xd).cpu().numpy() - refq).max()
    print("dead %-6s exact vs fp32 %.2e   kernels alone (plain) %.2e   conversion (plain, CPU) %.2e" % (name, e_exact, e_plain, np.abs(refq - ref).max()), flush=True)
