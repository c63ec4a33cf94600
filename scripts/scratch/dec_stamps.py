"""Tuning: phase durations of the beam-search step kernels (alt build with -DTN_DEC_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TENNIS_HIP_LIB"] = os.path.abspath("scripts/scratch/libs/stamps.so")   # hipcc -DTN_DEC_STAMPS build of captioner.hip linked with the other objects
import numpy as np, torch
from tennis_amd import weights as W, _lib
from tennis_amd.engine import GNMTCaptioner
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
p = W.make_gnmt_weights(0, "gru", F, H, E, V)
cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T)
src = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
vl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).to(dev)
for _ in range(2):
    cap.encode(src, vl); cap.beam_search(2, 3, 1.0, 5.0)
torch.cuda.synchronize()
lib = _lib.load()
out = (C.c_longlong * 24)()
lib.tn_dbg_dec_stamps.restype = C.c_int
lib.tn_dbg_dec_stamps(out)
st = np.array(list(out), dtype=np.int64)
names_a = ["gates0", "scores", "softmax", "context+write"]
names_b = [("gates1", 8, 9), ("proj", 9, 10), ("rows: logits+lse+cand+row top-k", 10, 12), ("merge top-k", 12, 13), ("bookkeeping", 13, 14), ("next inputs", 14, 16)]
print("valid_len[0] =", int(vl[0]))
print("attention (us):", {n: (st[i + 1] - st[i]) / 100.0 for i, n in enumerate(names_a)}, "total", (st[4] - st[0]) / 100.0)
print("beam (us):", {n: (st[b] - st[a]) / 100.0 for n, a, b in names_b}, "total", (st[16] - st[8]) / 100.0)
print("attention end -> beam start (lin1 + gaps):", (st[8] - st[4]) / 100.0)
try:
    o2 = (C.c_longlong * 8)()
    lib.tn_dbg_lat_stamps.restype = C.c_int
    lib.tn_dbg_lat_stamps(o2)
    l = np.array(list(o2), dtype=np.int64)
    print("gate GEMM workgroups (us after the attention kernel's last stamp): first %.2f..%.2f, middle %.2f..%.2f, last %.2f..%.2f; beam kernel's first stamp at %.2f" % (
        (l[0] - st[4]) / 100.0, (l[1] - st[4]) / 100.0, (l[2] - st[4]) / 100.0, (l[3] - st[4]) / 100.0, (l[4] - st[4]) / 100.0, (l[5] - st[4]) / 100.0, (st[8] - st[4]) / 100.0))
    print("first workgroup: %d s_memtime ticks in %.2f us -> %.0f MHz" % (l[7] - l[6], (l[1] - l[0]) / 100.0, (l[7] - l[6]) / ((l[1] - l[0]) / 100.0)))
except AttributeError:
    pass
