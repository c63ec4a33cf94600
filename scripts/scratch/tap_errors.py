"""Per-stage errors of the encoder against the numpy oracle for the default and the un-fused (TN_NO_FUSE) paths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
from oracle import densenet_np as dn
p = W.make_densenet121_weights(0)
x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(2, 224)).astype(np.float16)
taps = {}
ref = dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
xd = torch.from_numpy(x16.astype(np.float32)).cuda()
for mode in ("fused", "unfused"):
    if mode == "unfused": os.environ["TN_NO_FUSE"] = "1"
    enc = DenseNet121Features(p, 224, max_batch=2)
    feat = enc(xd).cpu().numpy()
    out = {"feat": float(np.abs(feat - ref).max())}
    for t in ("pool0", "stage1", "trans1", "stage2", "trans2", "stage3", "trans3", "stage4"):
        g = enc.read_tap(t, 2).reshape(taps[t].shape)
        d = np.abs(g - taps[t])
        out[t] = "%.2e/rms %.2e" % (d.max(), np.sqrt((d ** 2).mean()))
    print(mode, out)
    e = feat - ref
    print("   feature error: rms %.2e, correlation between the two frames' error vectors %.3f, mean %.2e" % (np.sqrt((e ** 2).mean()), np.corrcoef(e[0], e[1])[0, 1], e.mean()))
