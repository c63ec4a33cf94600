"""rocprofv3 target: the config-C3 temporal-head training step (32 clips x 64 steps x 1024 features), 10 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import TemporalHeadTrainer
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cell = os.environ.get("CELL", "gru")
B, T, F, H = 32, 64, 1024, 128
p = W.make_rnn_weights(0, cell, F, H, f"cnnrnn0_{cell}0_")
p.update(W.make_dense_weights(1, 11, 2 * H, "cnnrnn0_dense0_"))
x = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
y = torch.from_numpy(rng.integers(0, 11, B).astype(np.int32)).to(dev)
tr = TemporalHeadTrainer(p, F, H, 11, max_batch=B, max_steps=T, type=cell)
for _ in range(10):
    tr.forward_backward(x, y)
    tr.step(B, 1e-3, 0.9, 1e-4)
torch.cuda.synchronize()
