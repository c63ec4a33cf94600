"""`evaluate.py --save_feats` (reference evaluate.py:306-321): features of resident frames -> one `.npy` per frame.
Times the encode alone, the reference's np.save loop after each batch, and tennis_amd.evaluate.NpyWriter (tn_npy_writer_*:
a thread pool behind the C ABI that writes batch i while the GPU encodes batch i + 1).

    python scripts/bench_save_feats.py [--frames 256] [--batches 32] [--dir /tmp/feats] [--threads 16]
"""
import argparse, json, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--batches", type=int, default=32)
ap.add_argument("--dir", default="/tmp/tn_feats")
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()

from tennis_amd import weights as W
from tennis_amd.evaluate import NpyWriter
from tennis_amd.nn import DenseNet121Backbone

net = DenseNet121Backbone(seed=0)
x = torch.from_numpy(W.normalize_to_nchw_f32(W.synthetic_frames_u8(args.frames, 224))).cuda()
feat = net(x)
torch.cuda.synchronize()


def paths(tag, b):
    return [os.path.join(args.dir, tag, "V%03d" % b, "%05d.npy" % i) for i in range(args.frames)]


def run(tag, save, defer=False):
    """defer: the batch before is copied to the host and saved after this batch's forward has been queued (what
    tennis_amd.evaluate.save_features does since round 4)"""
    shutil.rmtree(os.path.join(args.dir, tag), ignore_errors=True)
    t0 = time.perf_counter()
    pending = None
    for b in range(args.batches):
        f = net(x)
        if save is not None:
            if defer:
                if pending is not None:
                    save(pending[0].cpu().numpy(), pending[1])
                pending = (f, paths(tag, b))
            else:
                save(f.cpu().numpy(), paths(tag, b))
    if pending is not None:
        save(pending[0].cpu().numpy(), pending[1])
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def np_save(host, ps):
    for i, p in enumerate(ps):
        if not os.path.exists(p):
            os.makedirs(os.path.dirname(p), exist_ok=True)
            np.save(p, host[i])


out = {"frames": args.frames, "batches": args.batches, "dir": args.dir, "cores": os.cpu_count()}
n = args.frames * args.batches
out["encode_only_fps"] = n / run("none", None)
out["np_save_loop_fps"] = n / run("py", np_save)
w = NpyWriter(threads=args.threads or None)
t0 = time.perf_counter()
dt = run("native", w.submit, defer=True)
written, skipped = w.drain()
dt_total = dt + 0.0
dt_total = time.perf_counter() - t0
out["npy_writer_fps"] = n / dt_total
out["npy_writer_submit_side_fps"] = n / dt
assert (written, skipped) == (n, 0)
a = open(paths("py", 3)[7], "rb").read()
b = open(paths("native", 3)[7], "rb").read()
assert a == b, "native file differs from np.save's"
w.close()
shutil.rmtree(args.dir, ignore_errors=True)
print(json.dumps(out))
