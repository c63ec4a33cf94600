"""(test infrastructure: imports oracle/, like everything under tests/)  The parity of the timed configuration on MORE frames than tests/test_gpu_calibration.py looks at (round 5): `--frames` frames
of each of the 16 families (other seeds than the test's and the calibration set's), features and Dense(11) logits of the default
calibrated conversion against the fp32 oracle (oracle/torch_ref.py) on the un-rounded weights and un-rounded input; also a second
weight seed.  Writes gpurun_out/parity_wide.json.     python tests/tools/parity_wide.py [--frames 8]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle.torch_ref import TorchDenseNet121
from tennis_amd import calib_frames as CF, weights as W
from tennis_amd.calibrate import calibrated_fp16_model
from tennis_amd.engine import DenseNet121Features

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=8); a = ap.parse_args()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_calibration import _jpeg_frames
torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
out = {"frames_per_family": a.frames, "bar": 1e-3, "weights": {}}
for wseed in (0, 1):
    p = W.make_densenet121_weights(wseed, fp16_model=False)
    wd = W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_")["framemodel0_dense0_weight"].astype(np.float64)
    q = calibrated_fp16_model(p, None, 224)
    enc = DenseNet121Features(q, 224, max_batch=a.frames)
    encx = DenseNet121Features(p, 224, max_batch=a.frames, exact_weights=True)       # hi + lo weight pairs everywhere: no conversion at all
    net = TorchDenseNet121(p)
    rows = {}
    for f in CF.FAMILIES + CF.HELD_OUT + ["jpeg"]:
        fr = CF.frames(f, a.frames, 224, seed=2024 + wseed) if f != "jpeg" else _jpeg_frames(a.frames + 3)[3:]
        ref = net(torch.from_numpy(W.normalize_to_nchw_f32(fr))).numpy()
        got = enc(torch.from_numpy(np.ascontiguousarray(fr)).cuda()).cpu().numpy()
        e = got.astype(np.float64) - ref
        ex = encx(torch.from_numpy(np.ascontiguousarray(fr)).cuda()).cpu().numpy().astype(np.float64) - ref
        rows[f] = {"feature_max": float(np.abs(e).max()), "feature_rms": float(np.sqrt((e ** 2).mean())), "logit_max": float(np.abs(e @ wd.T).max()),
                   "over_bar": int((np.abs(e) > 1e-3).sum()), "values": int(e.size),
                   "exact_mode_feature_max": float(np.abs(ex).max()), "exact_mode_over_bar": int((np.abs(ex) > 1e-3).sum())}
        print(wseed, f, rows[f], flush=True)
    out["weights"][f"seed {wseed}"] = {"families": rows, "worst_feature": max(r["feature_max"] for r in rows.values()),
                                       "worst_logit": max(r["logit_max"] for r in rows.values()), "values_over_bar": sum(r["over_bar"] for r in rows.values()),
                                       "exact_mode_worst_feature": max(r["exact_mode_feature_max"] for r in rows.values()),
                                       "exact_mode_values_over_bar": sum(r["exact_mode_over_bar"] for r in rows.values())}
    del enc, encx
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/parity_wide.json", "w"), indent=1)
print({k: (v["worst_feature"], v["worst_logit"], v["values_over_bar"], v["exact_mode_worst_feature"], v["exact_mode_values_over_bar"]) for k, v in out["weights"].items()})
