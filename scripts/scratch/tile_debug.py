"""Debug helper for dense_layer_big.hip geometries: error maps of one fused layer against the numpy reference."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tennis_amd import _lib
from oracle import densenet_np as dn
ctx = _lib.default_context(0)
B, H, K, ldc = (int(v) for v in sys.argv[1:5])
def _h(x): return x.astype(np.float16).astype(np.float32)
rng = np.random.default_rng(1)
buf = rng.normal(0, 1.5, (B, H, H, ldc)).astype(np.float16)
s1 = rng.uniform(0.5, 1.5, K).astype(np.float32); t1 = rng.normal(0, 0.3, K).astype(np.float32)
s2 = rng.uniform(0.5, 1.5, 128).astype(np.float32); t2 = rng.normal(0, 0.3, 128).astype(np.float32)
w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float16)
w3 = rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32)
wp = np.empty(2 * 72 * 64 * 8, np.uint16)
ctx.lib.tn_dbg_pack_conv3x3(w3.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(s1).cuda(), t1=torch.from_numpy(t1).cuda(), s2=torch.from_numpy(s2).cuda(),
         t2=torch.from_numpy(t2).cuda(), w1=torch.from_numpy(w1).cuda(), wp=torch.from_numpy(wp.view(np.int16)).cuda())
_lib.check(ctx.lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]), _lib.ptr(d["t1"]), _lib.ptr(d["w1"]),
                                          _lib.ptr(d["s2"]), _lib.ptr(d["t2"]), _lib.ptr(d["wp"]), B, H, H, None, 1), "dense_layer")
out = d["buf"].cpu().numpy().astype(np.float32)
a1 = _h(np.maximum(buf[..., :K].astype(np.float32) * s1 + t1, 0))
bott = (a1.reshape(-1, K) @ w1.astype(np.float32).T).reshape(B, H, H, 128)
a2 = _h(np.maximum(bott * s2 + t2, 0).astype(np.float32))
ref = dn.conv2d_nhwc(a2, _h(w3), 1, 1)
e = np.abs(out[..., K:K + 32] - ref)
np.set_printoptions(linewidth=250, precision=2, suppress=True)
print("max err", e.max()); print("by frame", e.max(axis=(1, 2, 3))); print("by row", e.max(axis=(0, 2, 3))); print("by col", e.max(axis=(0, 1, 3))); print("by ch", e.max(axis=(0, 1, 2)))
