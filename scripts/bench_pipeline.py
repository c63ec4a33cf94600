"""Files -> features: the reference's real-data route (dataset.py:204 imread -> evaluate.py:93-98 transform -> backbone) with
every stage on the GPU: JPEG decode (tn_jpeg_decode) -> Resize(256) / CenterCrop(224) (tn_preproc) -> DenseNet-121 features.
Reports frames/s of the three stages run one after the other, and with the decode of batch i+1 (its own stream, driven by a
second host thread) overlapping the transform + encode of batch i.

    python scripts/bench_pipeline.py [--frames 256] [--batches 8] [--height 720 --width 1280]
"""
import argparse, io, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--quality", type=int, default=90)
args = ap.parse_args()

from tennis_amd import _lib, image, transforms
from tennis_amd.nn import DenseNet121Backbone

rng = np.random.default_rng(0)
H, W = args.height, args.width
yy, xx = np.mgrid[0:H, 0:W]
files = []
for i in range(16):
    a = np.zeros((H, W, 3), np.float32)
    a[..., 1] = 110 + 30 * np.sin(xx / 200.0 + i); a[..., 0] = 60 + 20 * np.cos(yy / 150.0); a[..., 2] = 70
    a[(yy % 120 < 3) | (xx % 210 < 3)] = 235
    band = yy < H // 4
    a[band] = rng.integers(0, 255, (int(band.sum()), 3))
    a += rng.normal(0, 3, a.shape)
    b = io.BytesIO()
    Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", quality=args.quality, subsampling=2)
    files.append(b.getvalue())
batch = [files[i % 16] for i in range(args.frames)]

tf = transforms.Compose([transforms.Resize(256), transforms.CenterCrop(224), transforms.ToTensor(),
                         transforms.Normalize(transforms.IMAGENET_MEAN, transforms.IMAGENET_STD)])
net = DenseNet121Backbone(seed=0)
dec_stream = torch.cuda.Stream()            # the decoder works on its own stream: it may run beside the encoder
dec_ctx = _lib.Context(stream=dec_stream)
dec = image.JpegDecoder(dec_ctx)
bufs = [torch.empty((args.frames, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]

def encode(rgb):
    return net(tf(rgb))

feat = encode(dec.decode(batch, out=bufs[0]))
torch.cuda.synchronize()

t0 = time.perf_counter()
for i in range(args.batches):
    feat = encode(dec.decode(batch, out=bufs[i & 1]))
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / args.batches

# overlapped: a worker thread decodes batch i+1 while the main thread transforms + encodes batch i
t0 = time.perf_counter()
dec.decode(batch, out=bufs[0])
for i in range(args.batches):
    th = None
    if i + 1 < args.batches:
        th = threading.Thread(target=dec.decode, args=(batch,), kwargs=dict(out=bufs[(i + 1) & 1]))
        th.start()
    feat = encode(bufs[i & 1])
    torch.cuda.synchronize()
    if th is not None:
        th.join()
overl = (time.perf_counter() - t0) / args.batches
# two decoders on two streams, each driven by its own host thread, taking alternate batches: one decoder's host work (headers,
# staging copy, H2D) runs beside the other's kernels
import queue
dec2 = image.JpegDecoder(_lib.Context(stream=torch.cuda.Stream()))
bufs4 = bufs + [torch.empty_like(bufs[0]) for _ in range(2)]
def worker(d, first, q_out):
    for i in range(first, args.batches, 2):
        d.decode(batch, out=bufs4[i % 4])
        q_out.put(i)
t0 = time.perf_counter()
qs = [queue.Queue(), queue.Queue()]
ths = [threading.Thread(target=worker, args=(dec, 0, qs[0])), threading.Thread(target=worker, args=(dec2, 1, qs[1]))]
for th in ths: th.start()
for i in range(args.batches):
    assert qs[i & 1].get() == i
    feat = encode(bufs4[i % 4])
    torch.cuda.synchronize()      # (bufs4[i % 4] is free again only after its encode: 4 buffers, 2 decoders -> safe)
for th in ths: th.join()
two = (time.perf_counter() - t0) / args.batches
# N decoders, N streams, N host threads, and the main thread waits for ITS OWN stream only (an event behind the encode) - a
# device-wide synchronize after every encode, as above, also waits for the decoders' kernels of the next batches and puts the
# encoder in lock step with them
def many(nd, nb):
    decs = [dec, dec2] + [image.JpegDecoder(_lib.Context(stream=torch.cuda.Stream())) for _ in range(nd - 2)]
    free = [queue.Queue() for _ in range(nd)]
    done = [queue.Queue() for _ in range(nd)]
    for d in range(nd):
        for _ in range(2): free[d].put(torch.empty_like(bufs[0]))
    def work(d):
        for i in range(d, nb, nd):
            b = free[d].get()
            decs[d].decode(batch, out=b)
            done[d].put((i, b))
    ths = [threading.Thread(target=work, args=(d,)) for d in range(nd)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    pending = None
    for i in range(nb):
        j, b = done[i % nd].get()
        assert j == i
        f = encode(b)
        ev = torch.cuda.Event(); ev.record()
        if pending is not None:
            pending[0].synchronize(); free[pending[1]].put(pending[2])
        pending = (ev, i % nd, b)
    pending[0].synchronize()
    for th in ths: th.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nb
nb = max(args.batches, 12)
three = many(3, nb)
four = many(4, nb)
res = dict(two_decoders_ms_per_batch=round(two * 1e3, 2), two_decoders_frames_per_s=round(args.frames / two, 1), frames=args.frames, size=[H, W], serial_ms_per_batch=round(serial * 1e3, 2), serial_frames_per_s=round(args.frames / serial, 1),
           overlapped_ms_per_batch=round(overl * 1e3, 2), overlapped_frames_per_s=round(args.frames / overl, 1),
           three_decoders_own_stream_wait_ms_per_batch=round(three * 1e3, 2), three_decoders_own_stream_wait_frames_per_s=round(args.frames / three, 1),
           four_decoders_own_stream_wait_ms_per_batch=round(four * 1e3, 2), four_decoders_own_stream_wait_frames_per_s=round(args.frames / four, 1),
           decode_on_own_stream=True)
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_pipeline.json", "w"), indent=1)
