"""JPEG decode (reference dataset.py:204 ``mx.image.imread``): the CPU restatement against the real
decoder's golden vectors, and the host half of the C ABI (header parsing).  No GPU needed."""
import ctypes as C
import io
import os

import numpy as np
import pytest

from oracle import jpeg_np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz"))
CASES = sorted(k[:-6] for k in GOLD.files if k.endswith("__jpeg"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_libjpeg_golden(name):
    """the golden pairs are Pillow's (libjpeg-turbo's) own decodes: the oracle is pinned to them bit for bit"""
    out = jpeg_np.decode(GOLD[name + "__jpeg"].tobytes())
    assert out.dtype == np.uint8 and np.array_equal(out, GOLD[name + "__rgb"])


def test_oracle_against_live_pillow_matrix():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    for (h, w) in [(16, 16), (21, 35), (8, 8), (2, 2), (5, 3)]:
        for q in (35, 92):
            for ss in (0, 1, 2):
                a = (rng.random((h, w, 3)) * 255).astype(np.uint8)
                b = io.BytesIO()
                Image.fromarray(a).save(b, "JPEG", quality=q, subsampling=ss)
                ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                assert np.array_equal(jpeg_np.decode(b.getvalue()), ref), (h, w, q, ss)


def test_oracle_idct_dc_only_and_range_limit():
    """jidctint.c on a DC-only block is a flat block (DC * q / 8 rounded), and the range-limit table saturates"""
    q = np.ones(64, np.int32)
    for dc, want in ((0, 128), (8, 129), (-8, 127), (1016, 255), (2000, 255), (-1024, 0), (-2040, 0), (4, 129), (3, 128)):
        c = np.zeros((1, 64), np.int32)
        c[0, 0] = dc
        assert (jpeg_np.idct_islow(c, q) == want).all(), dc


def test_oracle_refuses_what_the_device_decoder_refuses():
    Image = pytest.importorskip("PIL.Image")
    a = (np.random.default_rng(0).random((32, 32, 3)) * 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", progressive=True)
    with pytest.raises(jpeg_np.JpegError, match="SOF2"):
        jpeg_np.decode(b.getvalue())
    with pytest.raises(jpeg_np.JpegError):
        jpeg_np.decode(b"\x00\x01\x02")


def test_abi_jpeg_info_on_host():
    """tn_jpeg_info walks the marker segments on the host: sizes, components, sampling; refusals carry a reason"""
    from tennis_amd import _lib
    lib = _lib.load()
    w, h, c, hs, vs = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    want_samp = {"c444": (1, 1), "c422": (2, 1), "c420": (2, 2), "grey": (1, 1)}
    for name in CASES:
        data = GOLD[name + "__jpeg"].tobytes()
        rc = lib.tn_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c), C.byref(hs), C.byref(vs))
        assert rc == 0, lib.tn_last_error()
        ref = GOLD[name + "__rgb"]
        assert (h.value, w.value) == ref.shape[:2]
        assert c.value == (1 if name.startswith("grey") else 3)
        assert (hs.value, vs.value) == want_samp[name[:4]]
    assert lib.tn_jpeg_info(b"\xff\xd8\xff\xe0\x00\x10JF", 8, None, None, None, None, None) == -1
    assert b"truncated" in lib.tn_last_error()
    assert lib.tn_jpeg_info(b"GIF89a....", 10, None, None, None, None, None) == -1
    assert b"SOI" in lib.tn_last_error()
    Image = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    assert lib.tn_jpeg_info(b.getvalue(), len(b.getvalue()), None, None, None, None, None) == -1
    assert b"SOF2" in lib.tn_last_error()
    from tennis_amd import image
    assert image.image_info(GOLD["c420_q85__jpeg"].tobytes()) == (53, 37, 3)
