"""rocprofv3 target: the fine-tuning step of the frame classifier, batch 16 at 224x224, 3 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import FrameModelTrainer
dev = torch.device("cuda:0")
p = W.make_densenet121_weights(0)
p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
B = int(os.environ.get("FT_BATCH", "16"))
tr = FrameModelTrainer(p, 224, 11, batch=B)
x = torch.randn((B, 224, 224, 3), device=dev)
y = torch.randint(0, 11, (B,), dtype=torch.int32, device=dev)
for _ in range(3):
    tr.forward_backward(x, y)
    tr.step(B, 1e-3, 0.9, 1e-4)
torch.cuda.synchronize()
