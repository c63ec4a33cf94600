"""-m gpu: each HIP kernel on its own against the CPU oracle (through the C ABI).

Kernel-level references emulate the kernel's two fp16 roundings (BN+ReLU output
and fp16 weights) so that indexing/layout bugs show up as O(1) errors while the
tolerance stays tight; the fp32-oracle comparison of the whole encoder lives in
test_gpu_encoder.py.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import densenet_np as dn
from oracle import rnn_np as rn
from oracle import vision_np as vn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tennis_amd import _lib
    return _lib.default_context(0)


def _h(x):  # round to fp16, back to fp32
    return x.astype(np.float16).astype(np.float32)


def _bnrelu_h(x, s, t):
    return _h(np.maximum(x * s + t, 0.0).astype(np.float32))


def _clamp_consts(rng, K):
    """(lo, hi) of a dense layer's BN1 + ReLU in the kernels' form (csrc/calib_host.hip::bn_relu_clamp_fold): the operand of the
    1x1 is clamp(x, lo, hi), fp16 numbers; most channels have a positive scale (lo = threshold, hi = 65504), some a negative one
    (lo = -65504, hi = threshold) and a few are constants (lo = hi = 0)"""
    thr = rng.normal(0, 0.6, K).astype(np.float16).astype(np.float32)
    kind = rng.random(K)
    lo = np.where(kind < 0.82, thr, np.float32(-65504.0)).astype(np.float32)
    hi = np.where(kind < 0.82, np.float32(65504.0), thr).astype(np.float32)
    lo[kind > 0.97] = 0.0; hi[kind > 0.97] = 0.0
    return lo, hi


def _clamp(x, lo, hi):
    return np.clip(x, lo, hi).astype(np.float32)       # exact: no arithmetic, no rounding


@pytest.mark.parametrize("M,K,ldx,N,yoff,ldy", [
    (1000, 64, 256, 128, 0, 128),      # ragged M, block-1 first layer
    (128 * 600, 96, 256, 128, 0, 128),  # BM=128 path, K tail (96 = 64+32)
    (64 * 600 + 7, 224, 512, 128, 0, 128),  # BM=64 path
    (300, 1024 - 32, 1024, 128, 8, 144),  # BM=32 path, output offset
])
def test_conv1x1(ctx, report, M, K, ldx, N, yoff, ldy):
    from tennis_amd import _lib
    rng = np.random.default_rng(M + K)
    x = rng.normal(0, 1.5, (M, ldx)).astype(np.float16)
    s = rng.uniform(0.5, 1.5, K).astype(np.float32)
    t = rng.normal(0, 0.3, K).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / K), (N, K)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((M, ldy), 7.0, dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.tn_dbg_conv1x1(ctx.handle, _lib.ptr(xd), ldx, K, s.ctypes.data_as(C.c_void_p),
                                      t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), N, _lib.ptr(yd),
                                      ldy, yoff, M, 0, 0, 0), "dbg_conv1x1")
    y = yd.cpu().numpy().astype(np.float32)
    ref = _bnrelu_h(x[:, :K].astype(np.float32), s, t) @ _h(w).T
    err = np.abs(y[:, yoff:yoff + N] - ref).max()
    report[f"conv1x1_M{M}_K{K}"] = float(err)
    assert err < 2e-2 * max(1.0, np.abs(ref).max() / 8), err
    # untouched columns keep their sentinel
    if ldy > N:
        mask = np.ones(ldy, bool); mask[yoff:yoff + N] = False
        assert np.all(y[:, mask] == 7.0)


@pytest.mark.parametrize("B,H,W,K,N", [(2, 28, 28, 512, 256), (3, 14, 14, 1024, 512), (1, 56, 56, 256, 128)])
def test_conv1x1_pool(ctx, report, B, H, W, K, N):
    from tennis_amd import _lib
    rng = np.random.default_rng(B * H + K)
    x = rng.normal(0, 1.5, (B, H, W, K)).astype(np.float16)
    s = rng.uniform(0.5, 1.5, K).astype(np.float32)
    t = rng.normal(0, 0.3, K).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / K), (N, K)).astype(np.float32)
    Mo = B * (H // 2) * (W // 2)
    ldy = N + 64
    xd = torch.from_numpy(x).cuda()
    yd = torch.zeros((Mo, ldy), dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.tn_dbg_conv1x1(ctx.handle, _lib.ptr(xd), K, K, s.ctypes.data_as(C.c_void_p),
                                      t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), N, _lib.ptr(yd),
                                      ldy, 0, Mo, 1, H, W), "dbg_conv1x1 pool")
    y = yd.cpu().numpy().astype(np.float32)[:, :N]
    a = np.maximum(x.astype(np.float32) * s + t, 0.0)
    a = _h(dn.avgpool(a, 2))
    ref = (a.reshape(-1, K) @ _h(w).T)
    err = np.abs(y - ref).max()
    report[f"conv1x1_pool_{B}x{H}x{W}x{K}"] = float(err)
    assert err < 2e-2, err


@pytest.mark.parametrize("B,H,W,K,N", [(5, 14, 14, 1024, 512), (3, 16, 16, 1024, 512), (2, 10, 12, 256, 512), (1, 14, 14, 128, 512),
                                       (2, 32, 32, 1024, 512), (3, 28, 28, 512, 256), (2, 64, 64, 512, 256), (1, 20, 22, 256, 256)])
def test_transition_warp_specialised_kernel_matches_the_tiled_one(ctx, B, H, W, K, N):
    """trans_ws.hip (round 6: the last transition with staging waves and multiplying waves, one frame per workgroup) against
    conv1x1.hip's tiled form on the same operands: the same operand tiles, the fp32 sums of 32x32x16 MFMAs instead of 16x16x32 ones -
    equal to one fp16 ulp, the columns outside the output untouched, two runs bit-identical (and the pooled map of a 16 x 16 /
    10 x 12 frame: 64 / 30 rows of the 64-row tile; the 128-pixel x 256-channel tile shape of the second transition: a 14 x 14 pooled
    map as two tiles of 98 rows, the maps of a 512 x 512 input as eight tiles of 128 / four of 64)."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * H + K)
    xd = torch.from_numpy(rng.normal(0, 1.5, (B, H, W, K)).astype(np.float16)).cuda()
    sd = torch.from_numpy(rng.uniform(0.5, 1.5, K).astype(np.float32)).cuda()
    td = torch.from_numpy(rng.normal(0, 0.3, K).astype(np.float32)).cuda()
    wd = torch.from_numpy(rng.normal(0, np.sqrt(2.0 / K), (N, K)).astype(np.float16)).cuda()
    Mo = B * (H // 2) * (W // 2)
    ldy = N + 64
    outs = []
    for variant in (0, 1 << 18, 1 << 18):
        yd = torch.full((Mo, ldy), 7.0, dtype=torch.float16, device="cuda")
        _lib.check(ctx.lib.tn_dbg_conv1x1_dev(ctx.handle, _lib.ptr(xd), K, K, _lib.ptr(sd), _lib.ptr(td), _lib.ptr(wd), N, _lib.ptr(yd),
                                              ldy, 32, Mo, 1, H, W, variant), "dbg_conv1x1_dev")
        torch.cuda.synchronize()
        outs.append(yd.cpu().numpy())
    assert np.isfinite(outs[0][:, 32:32 + N].astype(np.float32)).all() and np.abs(outs[0][:, 32:32 + N].astype(np.float32)).max() > 0.1
    assert np.array_equal(outs[1].view(np.uint16), outs[2].view(np.uint16))
    a, b = outs[0].astype(np.float32), outs[1].astype(np.float32)
    assert np.array_equal(a[:, :32], b[:, :32]) and np.array_equal(a[:, 32 + N:], b[:, 32 + N:])      # untouched
    d = np.abs(a - b)
    assert d.max() <= 2.0 ** -9 * max(1.0, np.abs(a).max()) and (d > 0).mean() < 0.05, (d.max(), (d > 0).mean(), np.abs(a).max())


@pytest.mark.parametrize("B,H,W", [(2, 14, 14), (3, 7, 7), (1, 56, 56), (2, 28, 28), (1, 9, 13)])
def test_conv3x3(ctx, report, B, H, W):
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 100 + H)
    x = rng.normal(0, 1.5, (B, H, W, 128)).astype(np.float16)
    s = rng.uniform(0.5, 1.5, 128).astype(np.float32)
    t = rng.normal(0, 0.3, 128).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32)
    ldy, yoff = 96, 40
    xd = torch.from_numpy(x).cuda()
    yd = torch.full((B * H * W, ldy), 3.0, dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.tn_dbg_conv3x3(ctx.handle, _lib.ptr(xd), s.ctypes.data_as(C.c_void_p),
                                      t.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), _lib.ptr(yd), ldy,
                                      yoff, B, H, W), "dbg_conv3x3")
    y = yd.cpu().numpy().astype(np.float32)
    a = _bnrelu_h(x.astype(np.float32), s, t)
    ref = dn.conv2d_nhwc(a, _h(w), 1, 1).reshape(-1, 32)
    err = np.abs(y[:, yoff:yoff + 32] - ref).max()
    report[f"conv3x3_{B}x{H}x{W}"] = float(err)
    assert err < 2e-2, err
    mask = np.ones(ldy, bool); mask[yoff:yoff + 32] = False
    assert np.all(y[:, mask] == 3.0)


@pytest.mark.parametrize("M,N,K", [(70, 11, 1024), (2048, 768, 1024), (33, 254, 356), (5, 7, 3),
                                   (4100, 520, 200), (4100, 513, 203), (160, 1024, 612)])   # 64x64-tile kernel (>= 512 tiles), scalar-load path, decoder cell
def test_linear(ctx, report, M, N, K):
    from tennis_amd import _lib
    rng = np.random.default_rng(M + N + K)
    x = rng.normal(0, 1, (M, K)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(K), (N, K)).astype(np.float32)
    b = rng.normal(0, 1, N).astype(np.float32)
    xd, wd, bd = (torch.from_numpy(a).cuda() for a in (x, w, b))
    yd = torch.empty((M, N), dtype=torch.float32, device="cuda")
    _lib.check(ctx.lib.tn_dbg_linear(ctx.handle, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(yd), M, N, K),
               "dbg_linear")
    ref = x.astype(np.float64) @ w.T.astype(np.float64) + b
    err = np.abs(yd.cpu().numpy() - ref).max()
    report[f"linear_{M}x{N}x{K}"] = float(err)
    assert err < 1e-4, err


@pytest.mark.parametrize("mode,B,T,F,H,use_vl", [("gru", 5, 9, 64, 128, False), ("lstm", 3, 7, 100, 128, False),
                                                  ("gru", 6, 11, 48, 256, True), ("lstm", 2, 5, 32, 64, True),
                                                  # the other shapes of the recurrent kernel: 1024-thread block (64
                                                  # weights in registers, the rest streamed), no register prefix,
                                                  # ragged tails (hidden % 16 != 0), four rows per workgroup
                                                  ("lstm", 2, 4, 16, 256, True), ("gru", 3, 5, 16, 32, False),
                                                  ("gru", 2, 4, 16, 100, True), ("lstm", 3, 4, 16, 84, False),
                                                  ("gru", 520, 3, 8, 32, True), ("lstm", 516, 2, 8, 36, False)])
def test_birnn(ctx, report, mode, B, T, F, H, use_vl):
    from tennis_amd import weights as Wt
    from tennis_amd.engine import BiRNN
    p = Wt.make_rnn_weights(3, mode, F, H, "rnn_")
    rng = np.random.default_rng(7)
    x = rng.normal(0, 1, (B, T, F)).astype(np.float32)
    vl = rng.integers(1, T + 1, B).astype(np.int32) if use_vl else None
    if vl is not None:
        vl[0] = T
    net = BiRNN(mode, F, H, p, "rnn_", True, max_rows=B * T, ctx=ctx)
    seq, hl, cl = net(torch.from_numpy(x).cuda(), None if vl is None else torch.from_numpy(vl).cuda(), True)
    ref, (fh, fc), (bh, bc) = rn.birnn_layer(x, p, "rnn_", mode, vl)
    err = np.abs(seq.cpu().numpy() - ref).max()
    errh = max(np.abs(hl[0].cpu().numpy() - fh).max(), np.abs(hl[1].cpu().numpy() - bh).max())
    report[f"birnn_{mode}_{B}x{T}x{F}x{H}_{use_vl}"] = float(max(err, errh))
    assert err < 1e-4 and errh < 1e-4, (err, errh)
    if mode == "lstm":
        errc = max(np.abs(cl[0].cpu().numpy() - fc).max(), np.abs(cl[1].cpu().numpy() - bc).max())
        assert errc < 1e-4, errc


def test_temporal_pool_and_prf1(ctx):
    from tennis_amd import _lib
    from tennis_amd.engine import temporal_pool
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (4, 9, 37)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    assert np.allclose(temporal_pool(xd, "max", ctx).cpu().numpy(), x.max(1))
    assert np.allclose(temporal_pool(xd, "mean", ctx).cpu().numpy(), x.mean(1), atol=1e-6)
    logits = rng.normal(0, 1, (513, 11)).astype(np.float32)
    logits[3, 2] = logits[3, 5] = 9.0  # tie -> first maximum
    labels = rng.integers(0, 11, 513).astype(np.int32)
    mat = torch.zeros((11, 11), dtype=torch.int64, device="cuda")
    ld, lab = torch.from_numpy(logits).cuda(), torch.from_numpy(labels).cuda()  # keep alive across the launch
    _lib.check(ctx.lib.tn_prf1_update(ctx.handle, _lib.ptr(ld), _lib.ptr(lab), 513, 11, _lib.ptr(mat)), "prf1")
    m = vn.PRF1([str(i) for i in range(11)])
    m.update([labels], [logits])
    assert np.array_equal(mat.cpu().numpy(), m.mat.astype(np.int64))


@pytest.mark.parametrize("B,H,K,ldc", [(2, 56, 64, 256), (1, 56, 224, 256), (3, 28, 128, 512), (2, 28, 480, 512),
                                        (3, 14, 256, 1024), (2, 14, 992, 1024), (8, 7, 512, 1024), (5, 7, 992, 1024), (3, 7, 544, 1024),
                                        (8, 28, 160, 512), (2, 14, 288, 1024), (1, 56, 96, 256),
                                        (33, 56, 64, 256), (40, 56, 96, 256),
                                        (2, 32, 256, 1024), (3, 32, 992, 1024), (70, 32, 416, 1024), (2, 16, 512, 1024), (5, 16, 992, 1024), (2, 64, 352, 512), (20, 64, 480, 512)])   # 32 / 16: the maps of a 512 x 512 input; > 256 tiles: persistent workgroups, uneven tile counts, with and without the XCD remap
@pytest.mark.parametrize("variant", [1, 9])   # K loops: 1 refill spread over the MFMA groups (default), 9 refill up front
def test_dense_layer_fused(ctx, report, B, H, K, ldc, variant):
    """One fused dense layer (1x1 -> LDS bottleneck tile -> 3x3, in-place concat) vs the oracle."""
    from tennis_amd import _lib
    W_ = H
    rng = np.random.default_rng(B * 1000 + H + K)
    buf = rng.normal(0, 1.5, (B, H, W_, ldc)).astype(np.float16)
    s1, t1 = _clamp_consts(rng, K)      # (lo, hi)
    s2 = rng.uniform(0.5, 1.5, 128).astype(np.float32); t2 = rng.normal(0, 0.3, 128).astype(np.float32)
    w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float16)
    w3 = rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32)
    wp = np.empty(2 * 72 * 64 * 8, np.uint16)   # both MFMA operand layouts
    ctx.lib.tn_dbg_pack_conv3x3(w3.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
    d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(s1).cuda(), t1=torch.from_numpy(t1).cuda(),
             s2=torch.from_numpy(s2).cuda(), t2=torch.from_numpy(t2).cuda(), w1=torch.from_numpy(w1).cuda(),
             wp=torch.from_numpy(wp.view(np.int16)).cuda())
    _lib.check(ctx.lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]),
                                              _lib.ptr(d["t1"]), _lib.ptr(d["w1"]), _lib.ptr(d["s2"]),
                                              _lib.ptr(d["t2"]), _lib.ptr(d["wp"]), B, H, W_, None, variant), "dense_layer")
    out = d["buf"].cpu().numpy().astype(np.float32)
    a1 = _clamp(buf[..., :K].astype(np.float32), s1, t1)
    bott = (a1.reshape(-1, K) @ w1.astype(np.float32).T).reshape(B, H, W_, 128)
    a2 = _h(np.maximum(bott * s2 + t2, 0).astype(np.float32))
    ref = dn.conv2d_nhwc(a2, _h(w3), 1, 1)
    err = np.abs(out[..., K:K + 32] - ref).max()
    report[f"dense_layer_fused_v{variant}_{B}x{H}_K{K}"] = float(err)
    assert err < 2e-2, err
    keep = np.ones(ldc, bool); keep[K:K + 32] = False
    assert np.array_equal(out[..., keep], buf[..., keep].astype(np.float32))


def _split_hi_lo_rows(w, kp):
    """[rows][2 kp] fp16 = [hi | lo] (+64 halves of slack), the exact-weights operand of the 1x1 convolutions"""
    n, k = w.shape
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    out = np.zeros((n, 2 * kp), np.float16)
    out[:, :k] = hi
    out[:, kp:kp + k] = lo
    return np.concatenate([out.ravel(), np.zeros(64, np.float16)])


@pytest.mark.parametrize("B,H,K,ldc", [(2, 56, 64, 256), (1, 56, 224, 256), (40, 56, 96, 256), (2, 28, 480, 512), (3, 28, 160, 512),
                                        (2, 14, 256, 1024), (2, 14, 288, 1024), (2, 14, 992, 1024), (3, 7, 512, 1024), (3, 7, 544, 1024)])
def test_dense_layer_exact_weights(ctx, report, B, H, K, ldc):
    """TN_ENC_EXACT_WEIGHTS in the fused dense layer (hi + lo fp16 pairs, K loop over [hi | lo], 18-tap phase B) vs the
    layer evaluated with the fp32 weights; K % 64 == 32 puts a dead k-step in the middle of the 64-channel loops."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 1000 + H + K)
    buf = rng.normal(0, 1.5, (B, H, H, ldc)).astype(np.float16)
    s1, t1 = _clamp_consts(rng, K)      # (lo, hi)
    s2 = rng.uniform(0.5, 1.5, 128).astype(np.float32); t2 = rng.normal(0, 0.3, 128).astype(np.float32)
    w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32)
    w3 = rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32)
    bk = 32 if H >= 28 else 64
    kp = (K + bk - 1) // bk * bk
    w3hi = _h(w3)
    imgs = []
    for part in (w3hi, w3 - w3hi):
        wp = np.empty(2 * 72 * 64 * 8, np.uint16)
        ctx.lib.tn_dbg_pack_conv3x3(np.ascontiguousarray(part, np.float32).ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
        imgs.append(wp)
    d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(s1).cuda(), t1=torch.from_numpy(t1).cuda(),
             s2=torch.from_numpy(s2).cuda(), t2=torch.from_numpy(t2).cuda(), w1=torch.from_numpy(_split_hi_lo_rows(w1, kp)).cuda(),
             wp=torch.from_numpy(np.concatenate(imgs).view(np.int16)).cuda())
    _lib.check(ctx.lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]), _lib.ptr(d["t1"]), _lib.ptr(d["w1"]),
                                              _lib.ptr(d["s2"]), _lib.ptr(d["t2"]), _lib.ptr(d["wp"]), B, H, H, None, 1 << 17), "dense_layer")
    out = d["buf"].cpu().numpy().astype(np.float32)
    a1 = _clamp(buf[..., :K].astype(np.float32), s1, t1)
    bott = (a1.reshape(-1, K).astype(np.float64) @ w1.astype(np.float64).T).reshape(B, H, H, 128).astype(np.float32)
    a2 = _h(np.maximum(bott * s2 + t2, 0).astype(np.float32))
    ref = dn.conv2d_nhwc(a2, w3, 1, 1)                                      # fp32 weights, un-rounded
    err = np.abs(out[..., K:K + 32] - ref).max()
    report[f"dense_layer_exact_{B}x{H}_K{K}"] = float(err)
    assert err < 6e-3, err            # fp16 output rounding (values up to ~8: half an ulp = 2e-3 .. 4e-3)
    keep = np.ones(ldc, bool); keep[K:K + 32] = False
    assert np.array_equal(out[..., keep], buf[..., keep].astype(np.float32))


@pytest.mark.parametrize("B,H,K,N", [(2, 56, 256, 128), (2, 28, 512, 256), (3, 14, 1024, 512)])
def test_transition_exact_weights(ctx, report, B, H, K, N):
    """TN_ENC_EXACT_WEIGHTS in the transition kernel (BN+ReLU, 2x2 average, 1x1 conv over [hi | lo] weights)."""
    from tennis_amd import _lib
    rng = np.random.default_rng(H + K)
    M = B * H * H
    x = rng.normal(0, 1.5, (M, K)).astype(np.float16)
    sc = rng.uniform(0.5, 1.5, K).astype(np.float32); sh = rng.normal(0, 0.3, K).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / K), (N, K)).astype(np.float32)
    Mo = M // 4
    y = torch.zeros((Mo, N), device="cuda", dtype=torch.float16)
    d = dict(x=torch.from_numpy(x).cuda(), sc=torch.from_numpy(sc).cuda(), sh=torch.from_numpy(sh).cuda(),
             w=torch.from_numpy(_split_hi_lo_rows(w, K)).cuda())
    _lib.check(ctx.lib.tn_dbg_conv1x1_dev(ctx.handle, _lib.ptr(d["x"]), K, K, _lib.ptr(d["sc"]), _lib.ptr(d["sh"]), _lib.ptr(d["w"]), N,
                                          _lib.ptr(y), N, 0, Mo, 1, H, H, 1 << 17), "conv1x1")
    a = np.maximum(x.astype(np.float32) * sc + sh, 0).reshape(B, H // 2, 2, H // 2, 2, K).mean(axis=(2, 4))
    a = _h(a).reshape(Mo, K)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    got = y.cpu().numpy().astype(np.float64)
    err = np.abs(got - ref).max()
    report[f"transition_exact_{H}_K{K}"] = float(err)
    assert err < 4e-3, err            # fp16 output rounding


@pytest.mark.parametrize("B,K0,nl,ldc", [(3, 512, 16, 1024), (2, 576, 3, 768), (1, 512, 1, 1024), (5, 448, 5, 1024), (2, 480, 7, 768),
                                          (300, 512, 2, 1024)])
def test_dense_block7(ctx, report, B, K0, nl, ldc):
    """The LDS-resident 7x7 dense block (dense_block7.hip: one frame per workgroup, concat buffer in LDS, weights streamed
    global -> registers, K split over the four waves) vs the oracle, layer by layer; every k-loop remainder (steps mod ring depth)
    occurs in the 16-layer case."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 1000 + K0 + nl)
    buf = np.zeros((B, 7, 7, ldc), np.float16)
    buf[..., :K0] = rng.normal(0, 1.0, (B, 7, 7, K0)).astype(np.float16)
    buf[..., K0:] = 77.0                                                  # must be overwritten (or left alone past the block)
    Ks = [K0 + 32 * l for l in range(nl)]
    s1, t1 = map(list, zip(*[_clamp_consts(rng, K) for K in Ks]))      # (lo, hi) per layer
    s2 = rng.uniform(0.5, 1.5, (nl, 128)).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
    w1 = [rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32) for K in Ks]
    w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (nl, 32, 128, 3, 3)).astype(np.float32))
    cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
    w1_all, s1_all, t1_all = cat(w1), cat(s1), cat(t1)
    h = C.c_void_p()
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    _lib.check(ctx.lib.tn_dbg_block7_create(ctx.handle, K0, nl, vp(w1_all), vp(s1_all), vp(t1_all), vp(s2), vp(t2), vp(w3), C.byref(h)), "block7_create")
    try:
        d = torch.from_numpy(buf).cuda()
        _lib.check(ctx.lib.tn_dbg_block7_run(h, _lib.ptr(d), ldc, B), "block7_run")
        out = d.cpu().numpy().astype(np.float32)
    finally:
        ctx.lib.tn_dbg_block7_destroy(h)
    ref = buf.astype(np.float32)
    worst = 0.0
    for l, K in enumerate(Ks):
        a1 = _clamp(ref[..., :K], s1[l], t1[l])
        bott = (a1.reshape(-1, K) @ _h(w1[l] * s2[l][:, None]).T).reshape(B, 7, 7, 128)
        a2 = _h(np.maximum(bott + t2[l], 0).astype(np.float32))
        y = _h(dn.conv2d_nhwc(a2, w3[l], 1, 1))
        worst = max(worst, float(np.abs(out[..., K:K + 32] - y).max()))
        ref[..., K:K + 32] = out[..., K:K + 32]          # follow the device's own roundings into the next layer
    report[f"dense_block7_{B}_K{K0}_nl{nl}"] = worst
    assert worst < 2e-2, worst
    end = K0 + 32 * nl
    assert np.array_equal(out[..., :K0], buf[..., :K0].astype(np.float32)) and np.all(out[..., end:] == 77.0)


@pytest.mark.parametrize("B,K0,nl,ldc", [(2, 256, 1, 1024), (3, 256, 4, 1024), (2, 288, 3, 512), (1, 256, 24, 1024), (5, 512, 16, 1024),
                                          (300, 320, 2, 1024)])
def test_dense_block14(ctx, report, B, K0, nl, ldc):
    """The streamed 14x14 dense block (dense_block14.hip: one frame per workgroup, pixel-owning waves, all weights through an
    LDS-DMA ring, activation ring in registers across layer boundaries, newest channels forwarded in registers) vs the oracle,
    layer by layer: odd and even super-step counts (both ring parities), the 24-layer block of a 224 x 224 input, the 16-layer
    14 x 14 block of a 448 x 448 input, more workgroups than CUs."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 1000 + K0 + nl)
    buf = np.zeros((B, 14, 14, ldc), np.float16)
    buf[..., :K0] = rng.normal(0, 1.0, (B, 14, 14, K0)).astype(np.float16)
    buf[..., K0:] = 77.0                                                  # must be overwritten (or left alone past the block)
    Ks = [K0 + 32 * l for l in range(nl)]
    s1, t1 = map(list, zip(*[_clamp_consts(rng, K) for K in Ks]))      # (lo, hi) per layer
    s2 = rng.uniform(0.5, 1.5, (nl, 128)).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
    w1 = [rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32) for K in Ks]
    w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (nl, 32, 128, 3, 3)).astype(np.float32))
    cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
    w1_all, s1_all, t1_all = cat(w1), cat(s1), cat(t1)
    h = C.c_void_p()
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    _lib.check(ctx.lib.tn_dbg_block14_create(ctx.handle, K0, nl, vp(w1_all), vp(s1_all), vp(t1_all), vp(s2), vp(t2), vp(w3), C.byref(h)), "block14_create")
    try:
        d = torch.from_numpy(buf).cuda()
        _lib.check(ctx.lib.tn_dbg_block14_run(h, _lib.ptr(d), ldc, B), "block14_run")
        out = d.cpu().numpy().astype(np.float32)
        d2 = torch.from_numpy(buf).cuda()                                  # a second run on a fresh copy: same bits
        _lib.check(ctx.lib.tn_dbg_block14_run(h, _lib.ptr(d2), ldc, B), "block14_run")
        assert torch.equal(d2, d)
    finally:
        ctx.lib.tn_dbg_block14_destroy(h)
    ref = buf.astype(np.float32)
    worst = 0.0
    for l, K in enumerate(Ks):
        a1 = _clamp(ref[..., :K], s1[l], t1[l])
        bott = (a1.reshape(-1, K) @ _h(w1[l] * s2[l][:, None]).T).reshape(B, 14, 14, 128)
        a2 = _h(np.maximum(bott + t2[l], 0).astype(np.float32))
        y = _h(dn.conv2d_nhwc(a2, w3[l], 1, 1))
        e = float(np.abs(out[..., K:K + 32] - y).max())
        assert e < 2e-2, (l, K, e)
        worst = max(worst, e)
        ref[..., K:K + 32] = out[..., K:K + 32]          # follow the device's own roundings into the next layer
    report[f"dense_block14_{B}_K{K0}_nl{nl}"] = worst
    end = K0 + 32 * nl
    assert np.array_equal(out[..., :K0], buf[..., :K0].astype(np.float32)) and np.all(out[..., end:] == 77.0)


@pytest.mark.parametrize("B,K0,nl,ldc", [(2, 128, 1, 512), (3, 128, 3, 512), (2, 160, 2, 256), (1, 128, 12, 512), (2, 384, 4, 512),
                                          (300, 256, 2, 512)])
def test_dense_block28(ctx, report, B, K0, nl, ldc):
    """The streamed 28x28 dense block (dense_block28.hip: one frame per workgroup walked in four passes of eight rows, all weights
    through the LDS-DMA ring once per pass, rolling bottleneck tile, 3x3 output rows one above the pass' 1x1 rows) vs the oracle,
    layer by layer: two-super-step layers (K = 128: the ring wraps through two passes), odd and even super-step counts, half-empty
    last super-steps, the 12-layer block of a 224 x 224 input, K > 320 (what the strip kernel could not hold), more workgroups than CUs."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 1000 + K0 + nl)
    buf = np.zeros((B, 28, 28, ldc), np.float16)
    buf[..., :K0] = rng.normal(0, 1.0, (B, 28, 28, K0)).astype(np.float16)
    buf[..., K0:] = 77.0                                                  # must be overwritten (or left alone past the block)
    Ks = [K0 + 32 * l for l in range(nl)]
    s1, t1 = map(list, zip(*[_clamp_consts(rng, K) for K in Ks]))      # (lo, hi) per layer
    s2 = rng.uniform(0.5, 1.5, (nl, 128)).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
    w1 = [rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32) for K in Ks]
    w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (nl, 32, 128, 3, 3)).astype(np.float32))
    cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
    w1_all, s1_all, t1_all = cat(w1), cat(s1), cat(t1)
    h = C.c_void_p()
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    _lib.check(ctx.lib.tn_dbg_block28_create(ctx.handle, K0, nl, vp(w1_all), vp(s1_all), vp(t1_all), vp(s2), vp(t2), vp(w3), C.byref(h)), "block28_create")
    try:
        d = torch.from_numpy(buf).cuda()
        _lib.check(ctx.lib.tn_dbg_block28_run(h, _lib.ptr(d), ldc, B), "block28_run")
        out = d.cpu().numpy().astype(np.float32)
        d2 = torch.from_numpy(buf).cuda()                                  # a second run on a fresh copy: same bits
        _lib.check(ctx.lib.tn_dbg_block28_run(h, _lib.ptr(d2), ldc, B), "block28_run")
        assert torch.equal(d2, d)
    finally:
        ctx.lib.tn_dbg_block28_destroy(h)
    ref = buf.astype(np.float32)
    worst = 0.0
    for l, K in enumerate(Ks):
        a1 = _clamp(ref[..., :K], s1[l], t1[l])
        bott = (a1.reshape(-1, K) @ _h(w1[l] * s2[l][:, None]).T).reshape(B, 28, 28, 128)
        a2 = _h(np.maximum(bott + t2[l], 0).astype(np.float32))
        y = _h(dn.conv2d_nhwc(a2, w3[l], 1, 1))
        err = np.abs(out[..., K:K + 32] - y)
        e = float(err.max())
        assert e < 2e-2, (l, K, e, np.argwhere(err > 2e-2)[:8].tolist())
        worst = max(worst, e)
        ref[..., K:K + 32] = out[..., K:K + 32]          # follow the device's own roundings into the next layer
    report[f"dense_block28_{B}_K{K0}_nl{nl}"] = worst
    end = K0 + 32 * nl
    assert np.array_equal(out[..., :K0], buf[..., :K0].astype(np.float32)) and np.all(out[..., end:] == 77.0)


@pytest.mark.parametrize("B,H,K,ldc", [(2, 56, 64, 256), (1, 56, 224, 256), (3, 56, 96, 256), (2, 56, 160, 256),
                                        (3, 28, 128, 512), (2, 28, 320, 512), (5, 28, 160, 512), (1, 28, 288, 512),
                                        (2, 128, 64, 256), (1, 128, 224, 256), (3, 64, 128, 512), (2, 64, 320, 512)])   # 512 x 512 input: 5 / 3 workgroups per frame, partly empty last strip pair
def test_dense_strip(ctx, report, B, H, K, ldc):
    """The strip-streaming fused dense layer (dense_strip_impl.h: one frame per workgroup, weights resident in LDS, bottleneck
    window in registers through chained MFMA layouts, 3x3 columns combined by DPP shifts) vs the oracle."""
    from tennis_amd import _lib
    rng = np.random.default_rng(B * 1000 + H + K)
    buf = rng.normal(0, 1.5, (B, H, H, ldc)).astype(np.float16)
    s1, t1 = _clamp_consts(rng, K)      # (lo, hi)
    s2 = rng.uniform(0.5, 1.5, 128).astype(np.float32); t2 = rng.normal(0, 0.3, 128).astype(np.float32)
    w1 = rng.normal(0, np.sqrt(2.0 / K), (128, K)).astype(np.float32)
    w3 = _h(rng.normal(0, np.sqrt(2.0 / 1152), (32, 128, 3, 3)).astype(np.float32))
    w1s = np.empty((K + 16) * 128, np.uint16); w3s = np.empty(36864, np.uint16)
    _lib.check(ctx.lib.tn_dbg_pack_strip(w1.ctypes.data_as(C.c_void_p), K, s2.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p),
                                         w1s.ctypes.data_as(C.c_void_p), w3.ctypes.data_as(C.c_void_p), w3s.ctypes.data_as(C.c_void_p)), "pack_strip")
    d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(s1).cuda(), t1=torch.from_numpy(t1).cuda(),
             w1s=torch.from_numpy(w1s.view(np.int16)).cuda(), w3s=torch.from_numpy(w3s.view(np.int16)).cuda())
    _lib.check(ctx.lib.tn_dbg_dense_strip_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]), _lib.ptr(d["t1"]),
                                              _lib.ptr(d["w1s"]), _lib.ptr(d["w3s"]), B, H, H, None), "dense_strip")
    out = d["buf"].cpu().numpy().astype(np.float32)
    a1 = _clamp(buf[..., :K].astype(np.float32), s1, t1)
    bott = (a1.reshape(-1, K) @ _h(w1 * s2[:, None]).T).reshape(B, H, H, 128)      # BN2's scale is folded into the weights before the fp16 rounding
    a2 = _h(np.maximum(bott + t2, 0).astype(np.float32))
    ref = dn.conv2d_nhwc(a2, w3, 1, 1)
    err = np.abs(out[..., K:K + 32] - ref).max()
    report[f"dense_strip_{B}x{H}_K{K}"] = float(err)
    assert err < 2e-2, err
    keep = np.ones(ldc, bool); keep[K:K + 32] = False
    assert np.array_equal(out[..., keep], buf[..., keep].astype(np.float32))
