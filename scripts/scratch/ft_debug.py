import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import densenet_train_torch as dt
from tennis_amd import weights as W
from tennis_amd.engine import FrameModelTrainer
B = 2
p = W.make_densenet121_weights(0); p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
x = W.normalize_to_nchw_f32(W.synthetic_frames_u8(B, 224, 5)); y = np.random.default_rng(5).integers(0, 11, B).astype(np.int32)
tr = FrameModelTrainer(p, 224, 11, batch=B)
loss, logits = tr.forward_backward(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
rl, rlog, rg, rstats = dt.loss_and_grads(p, x, y)
errs = []
for k, g in rg.items():
    got = tr.get(k, gradient=True)
    errs.append((float(np.abs(got - g).max() / max(1e-6, np.abs(g).max())), k, float(np.abs(g).max())))
errs.sort(reverse=True)
for e in errs[:12]: print(e)
print("median", np.median([e[0] for e in errs]))
for st in ("conv0", "batchnorm0_", "stage1_", "stage2_", "stage3_", "stage4_", "dense0"):
    v = [e[0] for e in errs if st in e[1]]
    print(st, len(v), max(v), np.median(v))
d = {k: e for e, k, _ in errs}
for k in ["densenet0_batchnorm4_gamma", "densenet0_batchnorm4_beta", "densenet0_stage4_conv31_weight", "densenet0_stage4_batchnorm31_gamma",
          "densenet0_stage4_batchnorm31_beta", "densenet0_stage4_conv30_weight", "densenet0_stage4_batchnorm30_gamma", "densenet0_stage4_batchnorm30_beta",
          "densenet0_stage4_conv29_weight", "densenet0_stage4_conv28_weight"]:
    print(k, d[k])
k = "densenet0_batchnorm4_beta"
got = tr.get(k, gradient=True); ref = rg[k]
i = np.argsort(-np.abs(got - ref))[:6]
print("idx", i, "got", got[i], "ref", ref[i])
k = "densenet0_stage4_batchnorm30_beta"
got = tr.get(k, gradient=True); ref = rg[k]
i = np.argsort(-np.abs(got - ref))[:6]
print("idx", i, "got", got[i], "ref", ref[i], "n bad", int((np.abs(got - ref) > 1e-4 * np.abs(ref).max()).sum()), "of", got.size)
