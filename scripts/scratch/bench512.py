"""frames/s of the encoder at another input size (scripts/scratch/bench512.py [size] [batch])"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
enc = DenseNet121Features(W.make_densenet121_weights(0), size, max_batch=B)
x = torch.randn((B, size, size, 3), device="cuda").half()
for _ in range(3): enc(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): enc(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"size {size} batch {B}: {dt*1e3:.2f} ms per batch, {B/dt:.0f} frames/s, {B/dt*(size/224)**2:.0f} 224-equivalent frames/s")
prof = getattr(enc, "profile", None)
stats, _ = enc.profile(x)
stats, _ = enc.profile(x)
tot = sum(s["ms"] for s in stats)
for s in stats:
    print("  %-32s %3d launches %7.3f ms  %5.1f %%  %6.1f TFLOP/s" % (s["name"], s["launches"], s["ms"], 100 * s["ms"] / tot, s["flops"] / max(s["ms"], 1e-9) / 1e9))
print("  total %.3f ms" % tot)
