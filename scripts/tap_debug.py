"""(tuning tool; imports oracle-free torch graph of scripts/round_study.py)  Where does the encoder's error against the fp32 graph appear?
Compares the block buffers (read_tap stage1..4) of a forward with the torch graph's maps, per block and per channel group.
    python scripts/tap_debug.py --weights trained --mode exact|plain|calibrated --batch 8 --family noise"""
import argparse, os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tennis_amd import weights as W, calib_frames as CF
from tennis_amd.engine import DenseNet121Features

ap = argparse.ArgumentParser()
ap.add_argument("--weights", default="trained"); ap.add_argument("--mode", default="exact"); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--family", default="noise"); ap.add_argument("--knobs", default="", help="LOWRANK,ROWGAIN,GAMMA_SIGMA,DEAD,NEG of tests/tools/trained_like.py")
ap.add_argument("--brief", action="store_true")
a = ap.parse_args()
if a.weights == "trained":
    import tools.trained_like as TL
    if a.knobs:
        TL.LOWRANK, TL.ROWGAIN, TL.GAMMA_SIGMA, TL.DEAD, TL.NEG = [float(v) for v in a.knobs.split(",")]
    p = TL.make_trained_like_weights(0)
else:
    p = W.make_densenet121_weights(0, fp16_model=False)
frames = CF.frames(a.family, a.batch, 224, seed=3)
if a.mode == "exact":
    model, enc = p, DenseNet121Features(p, 224, max_batch=a.batch, exact_weights=True)
elif a.mode == "plain":
    model = W.as_fp16_model(p); enc = DenseNet121Features(model, 224, max_batch=a.batch)
else:
    from tennis_amd.calibrate import calibrated_fp16_model
    model = calibrated_fp16_model(p, None, 224); enc = DenseNet121Features(model, 224, max_batch=a.batch)
ref_model = p if a.mode != "plain" else model          # plain: kernels alone (oracle on the converted weights)
P = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in ref_model.items()}
pre = "densenet0_"
def bn(x, n): return F.batch_norm(x, P[n + "_running_mean"], P[n + "_running_var"], P[n + "_gamma"], P[n + "_beta"], False, 0.0, 1e-5)
maps = {}
with torch.no_grad():
    x = torch.from_numpy(W.normalize_to_nchw_f32(frames))
    x = F.max_pool2d(F.relu(bn(F.conv2d(x, P[pre + "conv0_weight"], stride=2, padding=3), pre + "batchnorm0")), 3, 2, 1)
    outer = 1
    for st, nl in enumerate((6, 12, 24, 16), 1):
        sp = f"{pre}stage{st}_"
        for li in range(nl):
            y = F.conv2d(F.relu(bn(x, f"{sp}batchnorm{2 * li}")), P[f"{sp}conv{2 * li}_weight"])
            y = F.conv2d(F.relu(bn(y, f"{sp}batchnorm{2 * li + 1}")), P[f"{sp}conv{2 * li + 1}_weight"], padding=1)
            x = torch.cat([x, y], 1)
        maps[f"stage{st}"] = x.permute(0, 2, 3, 1).numpy().copy()
        if st != 4:
            x = F.avg_pool2d(F.conv2d(F.relu(bn(x, f"{pre}batchnorm{outer}")), P[f"{pre}conv{outer}_weight"]), 2, 2)
            outer += 1
    feat_ref = F.avg_pool2d(F.relu(bn(x, f"{pre}batchnorm{outer}")), 7).flatten(1).numpy()
feat = enc(torch.from_numpy(frames).cuda()).cpu().numpy()
print("features: max err %.3e  |ref| max %.2f" % (np.abs(feat - feat_ref).max(), np.abs(feat_ref).max()))
if a.brief:
    sys.exit(0)
import ctypes as C
from tennis_amd._lib import check
for st in (1, 2, 3, 4):
    ref = maps[f"stage{st}"]
    buf = np.empty(ref.size, np.float32); n = C.c_size_t(0)
    check(enc.lib.tn_densenet121_read_tap(enc.handle, f"stage{st}".encode(), a.batch, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(n)), "read_tap")
    got = buf[:n.value].reshape(ref.shape)
    e = np.abs(got - ref)
    cin = [64, 128, 256, 512][st - 1]
    groups = [("in", 0, cin)] + [(f"L{l}", cin + 32 * l, cin + 32 * l + 32) for l in range((ref.shape[-1] - cin) // 32)]
    print(f"stage{st}: " + " ".join("%s %.1e/%.1f" % (g, e[..., c0:c1].max(), np.abs(ref[..., c0:c1]).max()) for g, c0, c1 in groups))
