"""Phase stamps of the fused stem kernel (experimental build -DTN_STEM_STAMPS): TENNIS_HIP_LIB=... python scripts/scratch/stem_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["TN_NO_SPLIT"] = "1"
from tennis_amd import _lib, weights as W
from tennis_amd.engine import DenseNet121Features
lib = _lib.load()
p = W.make_densenet121_weights(0)
enc = DenseNet121Features(p, 224, max_batch=256)
x = torch.randn(256, 224, 224, 3, device="cuda").half()
for _ in range(2): enc(x)
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
lib.tn_dbg_stem_stamps.argtypes = [C.c_void_p, C.c_int]
lib.tn_dbg_stem_stamps(None, 1)
enc(x); torch.cuda.synchronize()
lib.tn_dbg_stem_stamps(buf.ctypes.data, 0)
v = buf.reshape(4096, 16).astype(np.float64)
v = v[v[:, 7] > 0]; print("tiles per WG", v[:, 7].mean())
names = ["prologue", "conv", "barrier1", "commit", "pool", "barrier2", "total", "n", "loadwait"]
print("WGs with stamps", len(v))
m = v.mean(0)
for i, nm in enumerate(names): print(f"  {nm:9s} {m[i]:9.0f} ticks per WG" + ("" if i in (0, 6, 7) else f"  ({m[i] / 4:7.0f} per tile)"))
