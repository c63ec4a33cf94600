"""Frames/s of the device JPEG decoder (csrc/jpeg.hip) on a batch of 720p frames, next to Pillow (libjpeg-turbo) on the
box's host cores - the reference's route (mx.image.imread on DataLoader workers, dataset.py:204; train.py:101-102).

    python scripts/bench_jpeg.py [--frames 256] [--height 720 --width 1280] [--quality 90] [--iters 10]
"""
import argparse, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--quality", type=int, default=90)
ap.add_argument("--subsampling", type=int, default=2)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--distinct", type=int, default=16, help="distinct images (the batch cycles through them)")
ap.add_argument("--decoders", default="2,3", help="also: N decoders on N streams, each driven by its own host thread (one's staging, H2D copy and\n"
                "latency-bound synchronisation rounds beside the others' kernels); comma-separated list of N, '' to skip")
args = ap.parse_args()

from tennis_amd import image
rng = np.random.default_rng(0)
H, W = args.height, args.width
yy, xx = np.mgrid[0:H, 0:W]
files = []
for i in range(args.distinct):
    # court-like content: large flat regions, lines, a textured crowd band, sensor noise
    a = np.zeros((H, W, 3), np.float32)
    a[..., 1] = 110 + 30 * np.sin(xx / 200.0 + i)
    a[..., 0] = 60 + 20 * np.cos(yy / 150.0)
    a[..., 2] = 70
    a[(yy % 120 < 3) | (xx % 210 < 3)] = 235
    band = yy < H // 4
    a[band] = rng.integers(0, 255, (int(band.sum()), 3))
    a += rng.normal(0, 3, a.shape)
    b = io.BytesIO()
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=args.quality, subsampling=args.subsampling)
    files.append(b.getvalue())
batch = [files[i % len(files)] for i in range(args.frames)]
nbytes = sum(len(b) for b in batch)

dec = image.JpegDecoder()
out = dec.decode(batch)
ref = np.asarray(Image.open(io.BytesIO(batch[3])).convert("RGB"))
assert np.array_equal(out[3].cpu().numpy(), ref), "device decode differs from Pillow"
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    dec.decode(batch, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters

# N decoders, N streams, N host threads: whole batches in parallel - how a loader with several workers drives it
multi = {}
if args.decoders:
    import threading
    from tennis_amd import _lib
    for nd in [int(x) for x in args.decoders.split(",")]:
        decs = [image.JpegDecoder(_lib.Context(stream=torch.cuda.Stream())) for _ in range(nd)]
        outs = [torch.empty_like(out) for _ in range(nd)]
        for d, o in zip(decs, outs): d.decode(batch, out=o)
        torch.cuda.synchronize()
        per = args.iters
        def work(d, o):
            for _ in range(per): d.decode(batch, out=o)
        reps = []
        for _ in range(3):           # (a host thread that is descheduled during its staging copy costs a whole batch: median of three)
            ths = [threading.Thread(target=work, args=(d, o)) for d, o in zip(decs, outs)]
            t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / (per * nd))
        dtn = sorted(reps)[1]
        assert np.array_equal(outs[-1][3].cpu().numpy(), ref), "device decode differs from Pillow (concurrent decoders)"
        multi[str(nd)] = dict(ms_per_batch=round(dtn * 1e3, 3), frames_per_s=round(args.frames / dtn, 1), reps_ms=[round(r * 1e3, 3) for r in reps])
        del decs, outs

k = min(32, args.frames)
t0 = time.perf_counter()
for b in batch[:k]:
    np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
pil = (time.perf_counter() - t0) / k
res = dict(frames=args.frames, size=[H, W], quality=args.quality, subsampling=args.subsampling, jpeg_mb=round(nbytes / 1e6, 2),
           device_ms_per_batch=round(dt * 1e3, 3), device_frames_per_s=round(args.frames / dt, 1),
           device_compressed_gb_per_s=round(nbytes / dt / 1e9, 3), sync_passes=dec.sync_passes, concurrent_decoders=multi,
           pillow_ms_per_frame_one_core=round(pil * 1e3, 3), pillow_frames_per_s_one_core=round(1 / pil, 1), host_cores=os.cpu_count())
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_jpeg.json", "w"), indent=1)
