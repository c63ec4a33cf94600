"""Debug helper for dense_strip.hip: error maps of one layer against the numpy reference."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tennis_amd import _lib
from oracle import densenet_np as dn

ctx = _lib.default_context(0)
B, H, K, ldc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mode = sys.argv[5] if len(sys.argv) > 5 else "full"
def _h(x): return x.astype(np.float16).astype(np.float32)
rng = np.random.default_rng(1)
buf = rng.normal(0, 1.5, (B, H, H, ldc)).astype(np.float16)
s1 = rng.uniform(0.5, 1.5, K).astype(np.float32); t1 = rng.normal(0, 0.3, K).astype(np.float32)
s2 = rng.uniform(0.5, 1.5, 128).astype(np.float32); t2 = rng.normal(0, 0.3, 128).astype(np.float32)
w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32)
w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32))
if mode == "center":      # only the centre tap: y = W3[:, :, 1, 1] a2  -> tests everything but the shifts
    w3[:, :, 0, :] = 0; w3[:, :, 2, :] = 0; w3[:, :, :, 0] = 0; w3[:, :, :, 2] = 0
if mode == "dy":          # centre column only
    w3[:, :, :, 0] = 0; w3[:, :, :, 2] = 0
if mode == "dx":          # centre row only
    w3[:, :, 0, :] = 0; w3[:, :, 2, :] = 0
w1s = np.empty((K + 16) * 128, np.uint16); w3s = np.empty(36864, np.uint16)
_lib.check(ctx.lib.tn_dbg_pack_strip(w1.ctypes.data_as(C.c_void_p), K, s2.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p),
                                     w1s.ctypes.data_as(C.c_void_p), w3.ctypes.data_as(C.c_void_p), w3s.ctypes.data_as(C.c_void_p)), "pack")
d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(s1).cuda(), t1=torch.from_numpy(t1).cuda(),
         w1s=torch.from_numpy(w1s.view(np.int16)).cuda(), w3s=torch.from_numpy(w3s.view(np.int16)).cuda())
_lib.check(ctx.lib.tn_dbg_dense_strip_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]), _lib.ptr(d["t1"]),
                                          _lib.ptr(d["w1s"]), _lib.ptr(d["w3s"]), B, H, H, None), "strip")
out = d["buf"].cpu().numpy().astype(np.float32)
a1 = _h(np.maximum(buf[..., :K].astype(np.float32) * s1 + t1, 0))
bott = (a1.reshape(-1, K) @ _h(w1 * s2[:, None]).T).reshape(B, H, H, 128)
a2 = _h(np.maximum(bott + t2, 0).astype(np.float32))
ref = dn.conv2d_nhwc(a2, w3, 1, 1)
e = np.abs(out[..., K:K + 32] - ref)
print(mode, "max err", e.max(), "ref absmax", np.abs(ref).max())
np.set_printoptions(linewidth=250, precision=2, suppress=True)
print("err by frame:", e.max(axis=(1, 2, 3)))
print("err by row:", e.max(axis=(0, 2, 3)))
print("err by col:", e.max(axis=(0, 1, 3)))
print("err by channel:", e.max(axis=(0, 1, 2)))
keep = np.ones(ldc, bool); keep[K:K + 32] = False
print("untouched ok:", np.array_equal(out[..., keep], buf[..., keep].astype(np.float32)))
