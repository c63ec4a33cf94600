"""rocprofv3 / timing target: the captioner's training step at config C5's shapes (32 clips x T=214 x F=1024, H=256, 20-token captions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import GNMTTrainer
rng = np.random.default_rng(0)
cell = os.environ.get("CELL", "gru")
B, T, F, H, E, V, L = 32, 214, 1024, 256, 100, 254, 20
p = W.make_gnmt_weights(0, cell, F, H, E, V)
tr = GNMTTrainer(p, F, H, E, V, max_batch=B, max_src_len=T, max_tgt_len=L, cell_type=cell)
src = torch.from_numpy((np.abs(rng.normal(0, 1, (B, T, F))) * 0.5).astype(np.float32)).cuda()
svl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).cuda()
tgt = torch.from_numpy(rng.integers(4, V, (B, L)).astype(np.int32)).cuda()
tvl = torch.from_numpy(rng.integers(8, L + 1, B).astype(np.int32)).cuda()
for _ in range(2):
    tr.forward_backward(src, svl, tgt, tvl); tr.step(1e-3) if hasattr(tr, "step") else None
torch.cuda.synchronize()
n = int(os.environ.get("ITERS", 10))
t0 = time.perf_counter()
for _ in range(n):
    tr.forward_backward(src, svl, tgt, tvl); tr.step(1e-3) if hasattr(tr, "step") else None
torch.cuda.synchronize()
print("captioner training step (%s): %.2f ms" % (cell, (time.perf_counter() - t0) / n * 1e3))
