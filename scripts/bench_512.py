"""512 x 512 input (the reference's default data_shape: train.py:44, 4096-d features train.py:259) at batch 256: frames/s of the encoder
(pipelined forwards, as bench.py) and where the time goes (tn_densenet121_profile).   python scripts/bench_512.py [--batch 256] [--steps 10]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import DenseNet121Features
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--size", type=int, default=512)
ap.add_argument("--out", default="gpurun_out/bench_512.json")
a = ap.parse_args()
p = W.make_densenet121_weights(0)
enc = DenseNet121Features(p, a.size, max_batch=a.batch)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randint(0, 256, (a.batch, a.size, a.size, 3), generator=g, device="cuda", dtype=torch.uint8)
out = torch.empty((a.batch, enc.feature_dim), dtype=torch.float32, device="cuda")
enc.set_pipelined(True)
for _ in range(3): enc(x, out=out)
enc.join(0); torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.steps): enc(x, out=out)
    enc.join(0); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
dt = float(np.median(ts))
fps = a.batch * a.steps / dt
flop_frame = 5.666e9 * (a.size / 224.0) ** 2
enc.set_pipelined(False)
stats, _ = enc.profile(x); stats, _ = enc.profile(x)
res = {"size": a.size, "batch": a.batch, "frames_per_s": round(fps, 1), "ms_per_batch": round(dt / a.steps * 1e3, 3), "tflops": round(fps * flop_frame / 1e12, 1),
       "frac_of_fp16_mfma_peak": round(fps * flop_frame / 2.5e15, 4), "workspace_gb": round(enc.workspace_bytes / 1e9, 2),
       "families_ms_unsplit_profile_pass": {s["name"]: {"ms": round(s["ms"], 3), "launches": s["launches"], "mfma_frac": round(s["flops"] / max(s["ms"], 1e-9) / 1e9 / 2.5e6, 4) if s["flops"] else None} for s in stats}}
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump(res, open(a.out, "w"), indent=1)
print(json.dumps(res, indent=1))
