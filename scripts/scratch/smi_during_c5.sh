#!/bin/bash
# Tuning: clocks / power while the captioner's beam search runs (a few small workgroups, launches back to back)
python - <<'PY' > /dev/null 2>&1 &
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import GNMTCaptioner
dev = torch.device("cuda:0"); rng = np.random.default_rng(0)
B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
p = W.make_gnmt_weights(0, "gru", F, H, E, V)
cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T)
src = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
vl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).to(dev)
for _ in range(1200):
    cap.encode(src, vl); cap.beam_search(2, 3, 1.0, 5.0)
torch.cuda.synchronize()
PY
BP=$!
for i in $(seq 1 14); do
  echo "t=$i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
kill $BP 2>/dev/null; wait $BP 2>/dev/null
