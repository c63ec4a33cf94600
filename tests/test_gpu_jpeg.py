"""JPEG decode on the device (csrc/jpeg.hip) through the C ABI: bit-exact against libjpeg's own output (golden
fixtures and live Pillow) and against oracle/jpeg_np.py; reference call site dataset.py:204,216 (``mx.image.imread``)."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import jpeg_np

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
CASES = sorted(k[:-6] for k in GOLD.files if k.endswith("__jpeg"))


def _img(rng, h, w, kind="smooth"):
    if kind == "noise":
        return (rng.random((h, w, 3)) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 120 * np.sin(xx / 17.0 + yy / 31.0), 127 + 120 * np.cos(xx / 13.0 - yy / 19.0), (xx * 3 + yy * 5) % 256], -1)
    return np.clip(a + rng.normal(0, 5, (h, w, 3)), 0, 255).astype(np.uint8)


def _encode(a, **kw):
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 1 << 24)      # optimize / restart options need the whole file in one encoder buffer
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


def _pillow(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.mark.parametrize("name", CASES)
def test_device_decode_matches_libjpeg_golden(name):
    from tennis_amd import image
    data = GOLD[name + "__jpeg"].tobytes()
    out = image.imdecode(data).cpu().numpy()
    assert out.dtype == np.uint8 and out.shape == GOLD[name + "__rgb"].shape
    assert np.array_equal(out, GOLD[name + "__rgb"])
    assert np.array_equal(out, jpeg_np.decode(data))       # and the oracle says the same


def test_device_decode_matrix_against_pillow():
    """sizes around the MCU boundaries x chroma layouts x qualities, one file per call"""
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(11)
    for (h, w) in [(16, 16), (17, 33), (64, 48), (100, 131), (8, 8), (2, 2), (5, 3), (1, 1), (31, 250), (240, 7)]:
        for ss in (0, 1, 2):
            for q, kind in ((30, "smooth"), (95, "noise"), (100, "noise")):
                data = _encode(_img(rng, h, w, kind), quality=q, subsampling=ss)
                out = image.imdecode(data).cpu().numpy()
                assert np.array_equal(out, _pillow(data)), (h, w, ss, q)


def test_device_decode_batch_of_a_video_720p():
    """a batch of 720p frames, every frame with its own quality (quantisation tables) and every second one with its own
    optimised Huffman tables; the scans are ~1000 subsequences long, so the parallel decoder really has to synchronise"""
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(12)
    base = _img(rng, 720, 1280)
    files = []
    for i in range(6):
        a = np.roll(base, 37 * i, axis=1)
        a = np.clip(a.astype(np.int16) + rng.integers(-20, 20, a.shape), 0, 255).astype(np.uint8)
        files.append(_encode(a, quality=60 + 7 * i, subsampling=2, optimize=bool(i & 1)))
    dec = image.JpegDecoder()
    out = dec.decode(files).cpu().numpy()
    assert out.shape == (6, 720, 1280, 3)
    for i, f in enumerate(files):
        assert np.array_equal(out[i], _pillow(f)), i
    assert 1 <= dec.sync_passes <= 8
    # the same handle again with another geometry and chroma layout (workspace re-use)
    small = [_encode(_img(rng, 90, 120, "noise"), quality=80, subsampling=1) for _ in range(3)]
    out2 = dec.decode(small).cpu().numpy()
    for i, f in enumerate(small):
        assert np.array_equal(out2[i], _pillow(f))


def test_device_decode_restart_intervals_and_grey():
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(13)
    a = _img(rng, 200, 312)
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=7), dict(restart_marker_rows=1), dict(restart_marker_rows=3)):
        for ss in (0, 2):
            data = _encode(a, quality=85, subsampling=ss, **kw)
            assert jpeg_np.parse(data)["ri"] > 0
            assert np.array_equal(image.imdecode(data).cpu().numpy(), _pillow(data)), (kw, ss)
    g = _encode(a[:, :, 1], quality=70)
    out = image.imdecode(g).cpu().numpy()
    assert out.shape == (200, 312, 3) and np.array_equal(out, _pillow(g))


def test_device_decode_refusals():
    """what the decoder does not handle fails loudly and names the file; a good call afterwards still works"""
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(14)
    good = _encode(_img(rng, 64, 64), quality=80, subsampling=2)
    prog = _encode(_img(rng, 64, 64), quality=80, progressive=True)
    other = _encode(_img(rng, 64, 80), quality=80, subsampling=2)
    with pytest.raises(RuntimeError, match="file 1: unsupported JPEG process SOF2"):
        image.imdecode_batch([good, prog])
    with pytest.raises(RuntimeError, match="file 1 differs from file 0"):
        image.imdecode_batch([good, other])
    with pytest.raises(RuntimeError, match="SOI"):
        image.imdecode(b"not a jpeg at all")
    cut = good[: len(good) // 2]
    with pytest.raises(RuntimeError, match="corrupt|truncated"):
        image.imdecode(cut)
    scr = bytearray(good)
    h = jpeg_np.parse(good)
    start = good.index(h["scan"][:16])
    for i in range(start + 40, start + 80):
        scr[i] = (scr[i] * 7 + 13) & 0xFE            # garbage (no FF bytes) in the middle of the scan
    try:
        out = image.imdecode(bytes(scr)).cpu().numpy()   # either refused as corrupt, or decoded like libjpeg decodes garbage
        assert out.shape == (64, 64, 3)
    except RuntimeError as e:
        assert "corrupt" in str(e)
    assert np.array_equal(image.imdecode(good).cpu().numpy(), _pillow(good))
    with pytest.raises(NotImplementedError):
        image.imdecode(good, flag=0)


def test_jpeg_to_features_end_to_end(tmp_path):
    """files on disk -> device decode -> Resize/CenterCrop (tn_preproc) -> encoder: identical features to the host-decoded
    route of the same files (the reference's route: imread on the host, transform, network)"""
    pytest.importorskip("PIL")
    from tennis_amd import image, transforms
    from tennis_amd.nn import DenseNet121Backbone
    rng = np.random.default_rng(15)
    paths = []
    for i in range(4):
        p = tmp_path / f"{i:010d}.jpg"
        p.write_bytes(_encode(_img(rng, 360, 640), quality=88, subsampling=2))
        paths.append(str(p))
    t = transforms.Compose([transforms.Resize(256), transforms.CenterCrop(224), transforms.ToTensor(),
                            transforms.Normalize(transforms.IMAGENET_MEAN, transforms.IMAGENET_STD)])
    dev_frames = image.imread_batch(paths)
    host_frames = np.stack([_pillow(open(p, "rb").read()) for p in paths])
    assert np.array_equal(dev_frames.cpu().numpy(), host_frames)
    net = DenseNet121Backbone(seed=3)
    fa = net(t(dev_frames)).float().cpu().numpy()
    fb = net(t(host_frames)).float().cpu().numpy()
    assert np.array_equal(fa, fb)
    one = image.imread(paths[2]).cpu().numpy()
    assert np.array_equal(one, host_frames[2])


@pytest.mark.parametrize("window", [1, 3])
def test_tennisset_device_decode_route(tmp_path, window):
    """TennisSet(decode="device"): the DataLoader sends the batch's JPEG files to the GPU as bytes and gets the same
    transformed uint8 batch as the host-decoded (Pillow) route, frame windows included"""
    pytest.importorskip("PIL")
    from test_cpu_input_side import _write_dataset
    from tennis_amd import transforms
    from tennis_amd.dataset import DataLoader, TennisSet
    root = str(tmp_path / "data")
    _write_dataset(root, np.random.default_rng(9), n_frames=(10, 9), size=(90, 160))
    t = transforms.Compose([transforms.Resize(256), transforms.CenterCrop(224), transforms.ToTensor(),
                            transforms.Normalize(transforms.IMAGENET_MEAN, transforms.IMAGENET_STD)])
    kw = dict(root=root, split="test", split_id="02", balance=False, transform=t, window=window)
    host = list(DataLoader(TennisSet(decode="host", **kw), batch_size=4))
    dev = list(DataLoader(TennisSet(decode="device", **kw), batch_size=4))
    auto = list(DataLoader(TennisSet(decode="auto", **kw), batch_size=4))
    assert len(host) == len(dev) == len(auto) > 1
    for (a, la, ia), (b, lb, ib), (c, lc, ic) in zip(host, dev, auto):
        assert b.is_cuda and b.dtype == torch.uint8 and a.shape == b.shape
        assert a.shape[1:] == ((224, 224, 3) if window == 1 else (window, 224, 224, 3))
        assert torch.equal(a, b) and torch.equal(a, c)
        assert np.array_equal(la, lb) and np.array_equal(ia, ib) and np.array_equal(ia, ic)
    # worker threads (two decoders on two streams): the same batches in the same order
    for nw in (1, 2):
        thr = list(DataLoader(TennisSet(decode="device", **kw), batch_size=4, num_workers=nw))
        assert len(thr) == len(host)
        for (a, la, ia), (b, lb, ib) in zip(host, thr):
            assert torch.equal(a, b) and np.array_equal(la, lb) and np.array_equal(ia, ib)
    with pytest.raises(ValueError):
        TennisSet(decode="gpu", **kw)


def test_device_decode_random_sweep():
    """120 files of random size (1..300 x 1..300), content, quality, chroma layout, Huffman optimisation and restart
    interval, decoded in random groups: every one identical to libjpeg's output"""
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(2024)
    dec = image.JpegDecoder()
    for _ in range(40):
        h, w = int(rng.integers(1, 301)), int(rng.integers(1, 301))
        kw = dict(quality=int(rng.integers(5, 101)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(0, 2)))
        r = int(rng.integers(0, 4))
        if r == 1:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        elif r == 2:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        files = []
        for i in range(3):          # three files of this geometry in one call (restart interval included in "geometry")
            a = _img(rng, h, w, "noise" if rng.integers(0, 2) else "smooth")
            if rng.integers(0, 4) == 0:
                a[:] = int(rng.integers(0, 256))          # a flat image: every AC coefficient is zero
            files.append(_encode(a, **kw))
        out = dec.decode(files).cpu().numpy()
        for i, f in enumerate(files):
            assert np.array_equal(out[i], _pillow(f)), (h, w, kw, i)


def test_device_decode_stuffing_heavy_streams():
    """Round 5: the FF 00 stuffing is removed on the device before the Huffman passes.  Streams with MANY stuffed bytes
    (quality 100 noise: long codes, runs of one-bits), with restart markers between stuffed bytes, multi-megabyte scans
    (thousands of 256-byte subsequences per file, un-stuffed ends far from the stuffed ones) and every alignment of the
    scan's first byte - all identical to libjpeg's output, and the stuffed-reader route (TN_JPEG_NO_UNSTUFF) is not needed."""
    pytest.importorskip("PIL")
    from tennis_amd import image
    rng = np.random.default_rng(77)
    dec = image.JpegDecoder()
    worst = 0.0
    for (h, w, kw) in [(256, 320, dict(quality=100, subsampling=0)), (240, 352, dict(quality=100, subsampling=2, restart_marker_blocks=3)),
                       (1080, 1920, dict(quality=98, subsampling=2)), (64, 64, dict(quality=100, subsampling=1, restart_marker_rows=1)),
                       (333, 517, dict(quality=100, subsampling=0, optimize=True))]:
        files = []
        for i in range(3):
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            a[:: 7 + i] = 255                                         # saturated rows: long runs of one-bits in the DC / AC values
            f = _encode(a, **kw)
            files.append(f if i != 1 else f[:2] + b"\xff\xfe" + (4 + i).to_bytes(2, "big") + b"ab" + b"c" * i + f[2:])   # a COM segment shifts the scan's alignment
        scan = files[0][files[0].rfind(b"\xff\xda"):]
        worst = max(worst, scan.count(b"\xff\x00") / max(1, len(scan)))
        out = dec.decode(files).cpu().numpy()
        for i, f in enumerate(files):
            assert np.array_equal(out[i], _pillow(f)), (h, w, kw, i)
    assert worst > 0.003          # the streams really are stuffing-heavy (random data: 1 / 256 of the bytes)


def test_device_decode_with_a_poisoned_coefficient_array():
    """Round 5: the coefficient array is no longer cleared per call - whole blocks leave the write pass as eight rows, and only
    the blocks that straddle a subsequence boundary are cleared first (jpeg_zero_straddle_kernel).  With TN_JPEG_POISON=1 the
    array is filled with 0x55 before every decode: any coefficient that nobody writes or clears would show as a wrong pixel.
    (The flag is read once per process: a child process.)"""
    pytest.importorskip("PIL")
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import io, sys, numpy as np
        sys.path.insert(0, REPO_ROOT)
        from PIL import Image
        from tennis_amd import image
        rng = np.random.default_rng(5)
        dec = image.JpegDecoder()
        for (h, w, kw) in [(720, 1280, dict(quality=90, subsampling=2)), (97, 131, dict(quality=35, subsampling=1)),
                           (256, 256, dict(quality=100, subsampling=0)), (200, 333, dict(quality=75, subsampling=2, restart_marker_blocks=5))]:
            files = []
            for i in range(5):
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.stack([127 + 100 * np.sin(xx / 23.0 + i), 127 + 100 * np.cos(yy / 17.0), (xx + yy * 3) % 256], -1)
                a[: h // 3] = rng.integers(0, 256, (h // 3, w, 3))
                if i == 4: a[:] = 37                       # a flat frame: blocks of one DC value
                b = io.BytesIO(); Image.fromarray(np.clip(a, 0, 255).astype(np.uint8)).save(b, "JPEG", **kw); files.append(b.getvalue())
            for rep in range(2):                          # the second call finds the first one's coefficients under the poison
                out = dec.decode(files).cpu().numpy()
                for i, f in enumerate(files):
                    assert np.array_equal(out[i], np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))), (h, w, kw, i, rep)
        print("ok")
    """).replace("REPO_ROOT", repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TN_JPEG_POISON="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_evaluate_save_feats_from_jpeg_frames(tmp_path):
    """``evaluate.py --save_feats`` on a dataset of JPEG frames on disk (reference evaluate.py:306-321 over the frames of
    dataset.py:204): the driver's default route - device decode, three loader threads - writes the same .npy feature files as
    the host-decode route"""
    pytest.importorskip("PIL")
    from test_cpu_input_side import _write_dataset
    from tennis_amd import evaluate as ev
    roots = []
    for name, extra in (("dev", []), ("host", ["--decode", "host", "--num_workers", "0"])):
        root = str(tmp_path / name)
        _write_dataset(root, np.random.default_rng(21), n_frames=(10, 9), size=(96, 128))
        assert ev.main(["--root", root, "--model_id", "0006", "--save_feats", "--batch_size", "4", "--split", "test"] + extra) == 0
        roots.append(root)
    files = []
    for dp, _dn, fn in os.walk(os.path.join(roots[0], "features", "0006")):
        files += [os.path.relpath(os.path.join(dp, f), roots[0]) for f in fn if f.endswith(".npy")]
    assert len(files) == 19                                     # --save_feats covers every frame of the split's videos (10 + 9)
    for f in files:
        a, b = np.load(os.path.join(roots[0], f)), np.load(os.path.join(roots[1], f))
        assert a.shape == (1024,) and a.dtype == np.float32 and np.array_equal(a, b), f
