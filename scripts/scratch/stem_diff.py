"""where does the fused stem differ from the oracle? (2 frames, pool0 tap)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
from oracle import densenet_np as dn
p = W.make_densenet121_weights(0)
enc = DenseNet121Features(p, 224, max_batch=4)
x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(2, 224)).astype(np.float16)
taps = {}
dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()
enc(xd)
ref = taps["pool0"]
got = enc.read_tap("pool0", 2).reshape(ref.shape)
err = np.abs(got - ref)
print("shape", ref.shape, "max err", err.max())
bad = err > 0.05
print("bad elements", bad.sum(), "of", bad.size)
# layout of ref? find axes
idx = np.argwhere(bad)
for ax in range(idx.shape[1]):
    vals, cnt = np.unique(idx[:, ax], return_counts=True)
    print("axis", ax, "size", ref.shape[ax], "bad values", vals[:40], "counts", cnt[:40])
