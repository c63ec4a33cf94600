"""rocprofv3 target: the config-C5 captioner only (encode + 150 beam-search steps), 3 iterations."""
import os, sys
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
for _ in range(3):
    cap.encode(src, vl)
    cap.beam_search(2, 3, 1.0, 5.0)
torch.cuda.synchronize()
