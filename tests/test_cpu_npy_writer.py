"""The ``.npy`` writer of ``evaluate.py --save_feats`` (csrc/npy_host.hip, reference evaluate.py:306-321): pure host code behind
the C ABI, checked against np.save byte for byte."""
import ctypes as C
import os

import numpy as np
import pytest

from tennis_amd import _lib
from tennis_amd.evaluate import NpyWriter


def test_files_are_np_save_bytes_and_existing_files_are_kept(tmp_path):
    rng = np.random.default_rng(5)
    w = NpyWriter(threads=4)
    for dim in (1, 7, 1024, 100003):            # header padding changes with the number of digits of the shape
        rows = rng.standard_normal((5, dim)).astype(np.float32)
        paths = [str(tmp_path / f"d{dim}" / f"v{i % 2}" / f"{i:05d}.npy") for i in range(5)]      # directories are created
        w.submit(rows, paths)
        assert w.drain() == (5, 0)
        for i, p in enumerate(paths):
            ref = tmp_path / "ref.npy"
            np.save(ref, rows[i])
            assert open(p, "rb").read() == open(ref, "rb").read()
            np.testing.assert_array_equal(np.load(p), rows[i])
    # evaluate.py:312: a file that exists is not rewritten
    rows2 = rng.standard_normal((5, 100003)).astype(np.float32)
    more = paths[:3] + [str(tmp_path / "new" / "a.npy"), str(tmp_path / "new" / "b.npy")]
    w.submit(rows2, more)
    assert w.drain() == (2, 3)
    np.testing.assert_array_equal(np.load(paths[0]), rows[0])
    np.testing.assert_array_equal(np.load(more[3]), rows2[3])
    # skip_existing=False overwrites
    w.submit(rows2[:1], paths[:1], skip_existing=False)
    assert w.drain() == (1, 0)
    np.testing.assert_array_equal(np.load(paths[0]), rows2[0])
    w.close()


def test_many_batches_in_flight(tmp_path):
    rng = np.random.default_rng(6)
    w = NpyWriter(threads=8)
    feats = rng.standard_normal((40, 64, 1024)).astype(np.float32)
    for b in range(40):                           # more jobs than the queue bound: submit blocks instead of piling up
        w.submit(feats[b], [str(tmp_path / f"{b:03d}" / f"{i:03d}.npy") for i in range(64)])
    assert w.drain() == (40 * 64, 0)
    for b in (0, 17, 39):
        for i in (0, 63):
            np.testing.assert_array_equal(np.load(tmp_path / f"{b:03d}" / f"{i:03d}.npy"), feats[b, i])
    w.close()


def test_errors_surface_at_drain(tmp_path):
    blocker = tmp_path / "file"
    blocker.write_bytes(b"x")
    w = NpyWriter(threads=2)
    w.submit(np.zeros((1, 4), np.float32), [str(blocker / "sub" / "a.npy")])     # a directory below a regular file
    with pytest.raises(RuntimeError, match="cannot create the directory"):
        w.drain()
    w.submit(np.ones((1, 4), np.float32), [str(tmp_path / "ok.npy")])            # the writer stays usable
    assert w.drain() == (1, 0)
    lib = _lib.load()
    assert lib.tn_npy_writer_submit(w.handle, None, 1, 4, None, 1) != 0
    h = C.c_void_p()
    assert lib.tn_npy_writer_create(0, C.byref(h)) != 0
    w.close()


def test_a_file_appears_under_its_name_only_when_complete(tmp_path):
    """the writer publishes a file by renaming a finished temporary, so a failed or interrupted write cannot leave a
    truncated .npy that the next ``--save_feats`` run would skip as 'exists'.  No temporaries are left behind; a file the writer
    cannot finish (its temporary cannot be created: the name is taken by a directory) does not appear at all."""
    rng = np.random.default_rng(7)
    w = NpyWriter(threads=4)
    rows = rng.standard_normal((32, 257)).astype(np.float32)
    paths = [str(tmp_path / "a" / f"{i:03d}.npy") for i in range(32)]
    w.submit(rows, paths)
    w.submit(rows, paths)                      # the same files again while the first batch may still be in flight: kept, not rewritten
    written, skipped = w.drain()
    assert written + skipped == 64 and written >= 32
    assert sorted(os.listdir(tmp_path / "a")) == [f"{i:03d}.npy" for i in range(32)]       # no *.tmp.* left
    for i in (0, 31):
        np.testing.assert_array_equal(np.load(paths[i]), rows[i])
    ro = tmp_path / "ro"
    ro.mkdir()
    os.chmod(ro, 0o555)
    try:
        if os.access(ro, os.W_OK):             # (root ignores the mode bits: nothing to provoke then)
            return
        w.submit(rows[:1], [str(ro / "x.npy")])
        with pytest.raises(RuntimeError):
            w.drain()
        assert os.listdir(ro) == []
    finally:
        os.chmod(ro, 0o755)
        w.close()


def test_stale_temporaries_are_swept(tmp_path):
    """a killed run's "<frame>.npy.tmp.<pid>.<thread>" files are removed when they are old; young ones (a live writer's) stay"""
    import time
    from tennis_amd.evaluate import NpyWriter
    d = tmp_path / "features" / "m" / "V006"
    d.mkdir(parents=True)
    old, young, real = d / "0001.npy.tmp.123.456", d / "0002.npy.tmp.123.457", d / "0003.npy"
    for f in (old, young, real):
        f.write_bytes(b"x")
    t = time.time() - 3600
    os.utime(old, (t, t))
    assert NpyWriter.sweep_stale(str(tmp_path / "features")) == 1
    assert not old.exists() and young.exists() and real.exists()
