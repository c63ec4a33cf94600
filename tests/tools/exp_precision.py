"""Checker tool (imports oracle/, hence under tests/; not collected by pytest).  Dev experiment: encoder error vs fp32 oracle with (a) fp32 weights, (b) fp16-representable conv weights."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
from oracle import densenet_np as dn

frames = W.synthetic_frames_u8(2, 224)
x16 = W.normalize_to_nchw_f32(frames).astype(np.float16)
xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()
res = {}
for tag in ["fp32w", "fp16w"]:
    p = W.make_densenet121_weights(0)
    if tag == "fp16w":
        for k in p:
            if k.endswith("_weight"):
                p[k] = p[k].astype(np.float16).astype(np.float32)
    enc = DenseNet121Features(p, 224, max_batch=2)
    taps = {}
    ref = dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
    f = enc(xd).cpu().numpy()
    d = np.abs(f - ref)
    res[tag] = dict(feat_max=float(d.max()), feat_mean=float(d.mean()), feat_rms=float(np.sqrt((d**2).mean())))
    for t in ["stem", "stage1", "stage2", "stage3", "stage4"]:
        g = enc.read_tap(t, 2).reshape(taps[t].shape)
        e = g - taps[t]
        res[tag][t] = dict(max=float(np.abs(e).max()), rms=float(np.sqrt((e**2).mean())), ref_rms=float(np.sqrt((taps[t]**2).mean())))
    del enc
print(json.dumps(res, indent=1))
