"""Experiment: does running two half-batches on two HIP streams (offset in layer position)
beat one full batch?  Steady-state throughput over many steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tennis_amd import _lib, weights as W
from tennis_amd.engine import DenseNet121Features
import bench

p = W.make_densenet121_weights(0)
dev = torch.device("cuda", 0)
x = bench.make_frames(256, 224, 1234, dev)

def run_single(steps):
    ctx = _lib.Context(0)
    enc = DenseNet121Features(p, 224, max_batch=256, ctx=ctx)
    out = torch.empty((256, 1024), device=dev)
    for _ in range(3): enc(x, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): enc(x, out=out)
    torch.cuda.synchronize(); return 256 * steps / (time.perf_counter() - t0)

def run_multi(steps, nstream, offset_cycles=0):
    per = 256 // nstream
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    ctxs = [_lib.Context(0, stream=s) for s in streams]
    encs = [DenseNet121Features(p, 224, max_batch=per, ctx=c) for c in ctxs]
    outs = [torch.empty((per, 1024), device=dev) for _ in range(nstream)]
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(nstream)]
    torch.cuda.synchronize()
    def step():
        for i in range(nstream):
            with torch.cuda.stream(streams[i]):
                encs[i](xs[i], out=outs[i])
    for _ in range(3): step()
    torch.cuda.synchronize()
    if offset_cycles:
        for i in range(1, nstream):
            with torch.cuda.stream(streams[i]):
                torch.cuda._sleep(offset_cycles * i)
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); return 256 * steps / (time.perf_counter() - t0)

print("single stream  B=256      : %.0f frames/s" % run_single(20))
for n, off in ((2, 0), (2, 1000000), (2, 2000000), (2, 3000000), (2, 4000000)):
    print("%d streams x B=%d offset %d cycles: %.0f frames/s" % (n, 256 // n, off, run_multi(40, n, off)))
