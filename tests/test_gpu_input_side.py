"""-m gpu: Resize + CenterCrop on the GPU (tn_preproc_*) bit-exact against oracle/image_np.py, and the reference's
transform_test chain feeding the encoder from JPEG frames on disk."""
import os

import numpy as np
import pytest
import torch

from oracle import image_np as im

pytestmark = pytest.mark.gpu


def _chain(s):
    from tennis_amd import transforms as T
    return T.Compose([T.Resize(s + 32), T.CenterCrop(s), T.ToTensor(),
                      T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])


@pytest.mark.parametrize("h,w,s", [(720, 1280, 224),      # TenniSet frames -> the reference's default data_shape
                                   (512, 512, 224),       # exact 2x reduction: OpenCV's box-average substitution
                                   (100, 180, 224),       # enlarging
                                   (257, 255, 224),       # ~1:1, borders clamp
                                   (48, 64, 32),
                                   (1080, 1920, 512)])    # the 4096-d feature configuration (data_shape 512)
def test_resize_crop_bit_exact(h, w, s):
    rng = np.random.default_rng(h * 7 + w)
    frames = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    got = _chain(s)(frames)
    assert got.is_cuda and got.dtype == torch.uint8 and tuple(got.shape) == (3, s, s, 3)
    ref = np.stack([im.test_transform_u8(f, s) for f in frames])
    assert np.array_equal(got.cpu().numpy(), ref)


def test_shapes_and_errors():
    from tennis_amd import transforms as T
    tf = _chain(64)
    rng = np.random.default_rng(0)
    one = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    win = np.stack([one] * 6).reshape(2, 3, 90, 120, 3)
    a = tf(one)
    b = tf(win)
    assert tuple(a.shape) == (64, 64, 3) and tuple(b.shape) == (2, 3, 64, 64, 3)
    assert torch.equal(b[1, 2], a) and np.array_equal(a.cpu().numpy(), im.test_transform_u8(one, 64))
    assert torch.equal(tf(torch.from_numpy(one).cuda()), a)          # frames already resident in HBM
    with pytest.raises(ValueError):
        tf(one.astype(np.float32))
    with pytest.raises(NotImplementedError):
        T.Compose([T.Resize(64), T.ToTensor()])
    with pytest.raises(NotImplementedError):
        T.Compose([T.Resize(64), T.CenterCrop(96), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    with pytest.raises(NotImplementedError):
        T.Resize(64, keep_ratio=True)


def test_abi_errors():
    import ctypes as C
    from tennis_amd import _lib
    ctx = _lib.default_context()
    h = C.c_void_p()
    assert ctx.lib.tn_preproc_create(ctx.handle, 0, 10, 8, 8, C.byref(h)) != 0
    assert ctx.lib.tn_preproc_create(ctx.handle, 10, 10, 8, 9, C.byref(h)) != 0
    assert b"crop" in ctx.lib.tn_last_error()
    assert ctx.lib.tn_preproc_forward(None, None, 1, None) != 0
    assert ctx.lib.tn_preproc_destroy(None) == 0


def test_frames_on_disk_to_features(tmp_path, report):
    """JPEG frames on disk -> TennisSet -> DataLoader (one Resize+CenterCrop launch per batch) -> DenseNet-121
    features, against the oracle chain: decoded frame -> image oracle -> ToTensor/Normalize -> fp32 encoder."""
    from test_cpu_input_side import _write_dataset
    from oracle import densenet_np as dn
    from tennis_amd import weights as W
    from tennis_amd.dataset import DataLoader, TennisSet, default_transform
    from tennis_amd.model_zoo import get_model
    root = str(tmp_path / "data")
    _write_dataset(root, np.random.default_rng(8), n_frames=(10, 9), size=(90, 160))
    ts = TennisSet(root=root, split="test", split_id="02", balance=False, transform=_chain(224))
    loader = DataLoader(ts, batch_size=4)
    backbone = get_model("DenseNet121", pretrained=True, seed=0).features
    p = W.make_densenet121_weights(0)
    feats, seen = [], []
    for data, labels, idxs in loader:
        assert data.is_cuda and data.dtype == torch.uint8 and tuple(data.shape[1:]) == (224, 224, 3)
        feats.append(backbone(data).cpu().numpy())
        seen += [int(i) for i in idxs]
    feats = np.concatenate(feats)
    assert seen == list(range(len(ts))) and feats.shape == (len(ts), 1024)
    pick = [0, 5, len(ts) - 1]
    x = np.stack([default_transform(im.test_transform_u8(ts.frame_u8(*ts._samples[i][:2]), 224)) for i in pick])
    ref = dn.densenet121_features(x, p)
    e = float(np.abs(feats[pick] - ref).max())
    report["disk_frames_to_features_maxabs_err"] = e
    assert e < 5e-3, e          # u8 path: one extra fp16 rounding of the normalised pixel (as test_gpu_encoder's u8 case)


def test_train_transform_matches_oracle_bit_for_bit():
    """Round 4: the reference's train transform (train.py:125-139) on the device (tn_augment_forward: crop + cv::resize + flip, the
    image's mean grey, jitter operators in the drawn order + lighting) against oracle/image_np.py::augment_u8 fed the SAME drawn
    parameters - uint8, bit for bit: 720p sources, a 2x-reduction window (INTER_AREA fast path), every one of the 24 operator
    orders, whole-frame windows, flips."""
    import ctypes as C
    from oracle import image_np as im
    from tennis_amd import transforms as T
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:360, 0:640]
    frames = np.stack([np.clip(np.stack([100 + 60 * np.sin(xx / (30.0 + i) + i), 90 + 50 * np.cos(yy / 25.0), 60 + (xx + yy + 13 * i) % 120], -1)
                               + rng.normal(0, 12, (360, 640, 3)), 0, 255).astype(np.uint8) for i in range(26)])
    tf = T.Compose([T.RandomResizedCrop(112), T.RandomFlipLeftRight(), T.RandomColorJitter(0.4, 0.4, 0.4), T.RandomLighting(0.1),
                    T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])], seed=5)
    rec = tf.draw_params(len(frames), 360, 640)
    import itertools
    for i, perm in enumerate(itertools.permutations(range(4))):              # every order once
        rec[i].order = sum(o << (2 * k) for k, o in enumerate(perm))
    rec[24].x0, rec[24].y0, rec[24].cw, rec[24].ch = 200, 100, 224, 224      # exact 2x reduction
    rec[25].x0, rec[25].y0, rec[25].cw, rec[25].ch = 0, 0, 640, 360          # the whole frame
    got = tf.augment(torch.from_numpy(frames).cuda(), rec).cpu().numpy()
    for i in range(len(frames)):
        r = rec[i]
        want = im.augment_u8(frames[i], r.x0, r.y0, r.cw, r.ch, r.flip, [(r.order >> (2 * k)) & 3 for k in range(4)], r.brightness,
                             r.contrast, r.saturation, list(r.light), 112)
        assert np.array_equal(got[i], want), (i, int(np.abs(got[i].astype(int) - want.astype(int)).max()), (r.x0, r.y0, r.cw, r.ch, r.flip, r.order))
    # the call path: a batch through Compose.__call__ (its own draws), shape / dtype / reproducibility from the seed
    a = T.Compose([T.RandomResizedCrop(64), T.RandomFlipLeftRight(), T.RandomColorJitter(0.4, 0.4, 0.4), T.RandomLighting(0.1), T.ToTensor(),
                   T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])], seed=9)
    b = T.Compose([T.RandomResizedCrop(64), T.RandomFlipLeftRight(), T.RandomColorJitter(0.4, 0.4, 0.4), T.RandomLighting(0.1), T.ToTensor(),
                   T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])], seed=9)
    xa, xb = a(frames[:8]), b(frames[:8])
    assert xa.shape == (8, 64, 64, 3) and xa.dtype == torch.uint8 and torch.equal(xa, xb) and not torch.equal(a(frames[:8]), xa)
    # errors: a window outside the frame, a broken order
    bad = tf.draw_params(1, 360, 640)
    bad[0].cw = 700
    with pytest.raises(RuntimeError, match="crop window"):
        tf.augment(torch.from_numpy(frames[:1]).cuda(), bad)
    bad = tf.draw_params(1, 360, 640)
    bad[0].order = 0
    with pytest.raises(RuntimeError, match="permutation"):
        tf.augment(torch.from_numpy(frames[:1]).cuda(), bad)
