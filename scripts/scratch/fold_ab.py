"""A/B of the un-fused 1x1 -> 3x3 pipeline with BN2's scale applied in the 3x3 kernel vs folded into the 1x1 weights."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tennis_amd import _lib
from oracle import densenet_np as dn
ctx = _lib.default_context(0)
def _h(x): return x.astype(np.float16).astype(np.float32)
rng = np.random.default_rng(3)
B, H, K = 2, 28, 256
x = rng.normal(0, 1.5, (B * H * H, K)).astype(np.float16)
s1 = rng.uniform(0.5, 1.5, K).astype(np.float32); t1 = rng.normal(0, 0.3, K).astype(np.float32)
s2 = rng.uniform(0.73, 1.34, 128).astype(np.float32); t2 = rng.normal(0, 0.1, 128).astype(np.float32)
w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32)
w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32))
P = lambda a: a.ctypes.data_as(C.c_void_p)
def run(w1_used, s2_used):
    xd = torch.from_numpy(x).cuda()
    bott = torch.zeros((B * H * H, 128), dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.tn_dbg_conv1x1(ctx.handle, _lib.ptr(xd), K, K, P(s1), P(t1), P(w1_used), 128, _lib.ptr(bott), 128, 0, B * H * H, 0, 0, 0), "c1")
    y = torch.zeros((B * H * H, 32), dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.tn_dbg_conv3x3(ctx.handle, _lib.ptr(bott), P(s2_used), P(t2), P(w3), _lib.ptr(y), 32, 0, B, H, H), "c3")
    return y.cpu().numpy().astype(np.float64), bott.cpu().numpy().astype(np.float64)
a1 = _h(np.maximum(x.astype(np.float32) * s1 + t1, 0)).astype(np.float64)
for name, wq in (("unfolded", _h(w1)), ("folded", _h(w1 * s2[:, None]) / s2[:, None])):
    z = a1 @ wq.astype(np.float64).T
    a2 = np.maximum(z * s2 + t2, 0)
    ref = dn.conv2d_nhwc(a2.reshape(B, H, H, 128).astype(np.float32), w3, 1, 1).reshape(-1, 32).astype(np.float64)
    if name == "unfolded":
        y, bott = run(np.ascontiguousarray(_h(w1)), s2)
    else:
        y, bott = run(np.ascontiguousarray((w1 * s2[:, None]).astype(np.float32)), np.ones(128, np.float32))
    cm = (y - ref).mean(axis=0)
    zref = z * s2
    print("   channel-mean error of the output: rms %.3e max %.3e | bottleneck rounding error mean over pixels per channel rms %.3e" % (np.sqrt((cm ** 2).mean()), np.abs(cm).max(), np.sqrt(((bott - (zref if name == "folded" else z)).mean(axis=0) ** 2).mean())))
    print(name, "out max err %.3e rms %.3e" % (np.abs(y - ref).max(), np.sqrt(((y - ref) ** 2).mean())), "| bott rms", np.sqrt((bott ** 2).mean()))
