"""ad-hoc: encoder vs the torch-CPU restatement at assorted input sizes (H, W)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle.torch_ref import TorchDenseNet121
from tennis_amd import weights as W
from tennis_amd import _lib
import ctypes as C
p = W.make_densenet121_weights(0)
ref_model = TorchDenseNet121(p)
from tennis_amd.engine import DenseNet121Features
for (h, w) in [(224, 448), (448, 224), (256, 320), (224, 230), (672, 224), (896, 896)]:
    rng = np.random.default_rng(h * 7 + w)
    x = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    x16 = ((x.astype(np.float32) / 255.0 - W.IMAGENET_MEAN) / W.IMAGENET_STD).transpose(0, 3, 1, 2).astype(np.float16)
    try:
        enc = DenseNet121Features(p, (h, w), max_batch=1)
    except Exception as e:
        print((h, w), "create failed:", str(e)[:150]); continue
    got = enc(torch.from_numpy(x16.astype(np.float32)).cuda()).cpu().numpy()
    with torch.no_grad():
        ref = ref_model(torch.from_numpy(x16.astype(np.float32))).numpy()
    print((h, w), got.shape, ref.shape, "max|err| %.2e" % float(np.abs(got - ref.reshape(got.shape)).max()) if got.size == ref.size else "shape mismatch")
