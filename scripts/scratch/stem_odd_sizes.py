import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
from oracle import densenet_np as dn
p = W.make_densenet121_weights(0)
for size in (232, 236, 200, 226):
    try:
        frames = W.synthetic_frames_u8(1, size)
        x16 = W.normalize_to_nchw_f32(frames).astype(np.float16)
        taps = {}
        ref = dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
        enc = DenseNet121Features(p, size, max_batch=1)
        for name, x in (("nhwc16", torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()),
                        ("nchw32", torch.from_numpy(x16.astype(np.float32)).cuda()),
                        ("u8", torch.from_numpy(frames).cuda())):
            f = enc(x).cpu().numpy()
            got = enc.read_tap("pool0", 1).reshape(taps["pool0"].shape)
            print(size, name, "pool0 err", float(np.abs(got - taps["pool0"]).max()), "feat err", float(np.abs(f - ref).max()), f.shape)
    except Exception as e:
        print(size, "ERROR", repr(e)[:300])
