import sys, io, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_jpeg import _img, _encode, _pillow
from tennis_amd import image
rng = np.random.default_rng(2024)
dec = image.JpegDecoder()
for it in range(40):
    h, w = int(rng.integers(1, 301)), int(rng.integers(1, 301))
    kw = dict(quality=int(rng.integers(5, 101)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
    r = int(rng.integers(0, 4))
    if r == 1: kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    elif r == 2: kw["restart_marker_rows"] = int(rng.integers(1, 4))
    files = []
    for i in range(3):
        a = _img(rng, h, w, "noise" if rng.integers(0, 2) else "smooth")
        if rng.integers(0, 4) == 0: a[:] = int(rng.integers(0, 256))
        files.append(_encode(a, **kw))
    print(it, h, w, kw, [len(f) for f in files], flush=True)
    out = dec.decode(files).cpu().numpy()
    for i, f in enumerate(files):
        assert np.array_equal(out[i], _pillow(f)), (h, w, kw, i)
print("all ok")
