"""Wall clock of the config-C5 beam search (32 clips, beam 5, H = 256, T = 214, V = 254, 150 steps): µs per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import GNMTCaptioner
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cell = os.environ.get("CELL", "gru")
B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
p = W.make_gnmt_weights(0, cell, F, H, E, V)
cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T, cell_type=cell) if cell != "gru" else \
    GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T)
src = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
vl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).to(dev)
cap.encode(src, vl)
s, _, v = cap.beam_search(2, 3, 1.0, 5.0)
torch.cuda.synchronize()
width = s.shape[-1]
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    cap.beam_search(2, 3, 1.0, 5.0)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print("C5 %s beam search: %.2f ms, %d columns -> %.1f us per step" % (cell, best * 1e3, width, best * 1e6 / (width - 2)))
