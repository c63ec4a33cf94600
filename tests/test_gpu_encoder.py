"""-m gpu: the whole DenseNet-121 frame encoder through the C ABI vs the fp32 CPU oracle.

Tolerance (BASELINE.json north_star): features/logits within 1e-3 absolute of the
fp32 CPU path.  Inputs are fp16-representable so that input quantisation is not
charged to the kernels; per-stage errors are recorded in gpurun_out/parity_report.json.
"""
import numpy as np
import pytest
import torch

from oracle import densenet_np as dn

pytestmark = pytest.mark.gpu

TOL = 1e-3


@pytest.fixture(scope="module")
def setup():
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    enc = DenseNet121Features(p, 224, max_batch=4)
    frames = W.synthetic_frames_u8(2, 224)
    x32 = W.normalize_to_nchw_f32(frames)
    x16 = x32.astype(np.float16)          # NCHW fp16-representable values
    taps = {}
    ref = dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
    return dict(p=p, enc=enc, frames=frames, x16=x16, ref=ref, taps=taps)


def test_encoder_stages(setup, report):
    enc, x16 = setup["enc"], setup["x16"]
    xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()  # NHWC fp16
    feat = enc(xd).cpu().numpy()
    worst = {}
    for tap in ["pool0", "stage1", "trans1", "stage2", "trans2", "stage3", "trans3", "stage4"]:
        ref = setup["taps"][tap]
        got = enc.read_tap(tap, 2).reshape(ref.shape)
        e = float(np.abs(got - ref).max())
        worst[tap] = e
        report[f"encoder_tap_{tap}_maxabs"] = e
        report[f"encoder_tap_{tap}_refabsmax"] = float(np.abs(ref).max())
    e = float(np.abs(feat - setup["ref"]).max())
    report["encoder_feat_maxabs_err"] = e
    report["encoder_feat_mean_abs_err"] = float(np.abs(feat - setup["ref"]).mean())
    # stage activations are fp16 (half-ulp up to 4e-3 at |x|~8): loose per-stage sanity bound
    for tap, v in worst.items():
        assert v < 0.1, (tap, v)
    assert e < TOL, f"feature max abs err {e} vs fp32 oracle"


def test_encoder_layouts_agree(setup, report):
    """NCHW fp32 (reference layout), NHWC fp16 and NHWC u8 inputs give the same features."""
    enc = setup["enc"]
    x16 = setup["x16"]
    f_nhwc = enc(torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()).cpu().numpy()
    f_nchw = enc(torch.from_numpy(x16.astype(np.float32)).cuda()).cpu().numpy()
    assert np.array_equal(f_nhwc, f_nchw)
    f_u8 = enc(torch.from_numpy(setup["frames"]).cuda()).cpu().numpy()
    e = float(np.abs(f_u8 - setup["ref"]).max())
    report["encoder_u8_feat_maxabs_err"] = e
    assert e < TOL, e    # (the u8 path derives the normalised pixel in fp32 and rounds it once: measured 5.5e-4)


def test_unrounded_input_is_inside_the_bar(setup, report):
    """What the reference's test transform produces (evaluate.py:93-98: ToTensor + Normalize of the uint8 frame) is fp32 and
    NOT fp16-representable; the encoder rounds it to fp16 on the way in (fp32 NCHW hand-over: in the layout kernel; uint8
    hand-over: (u8/255 - mean)/std evaluated in fp32, rounded once).  Here the oracle is fed the UN-rounded fp32 tensor: the
    input rounding is part of the measured error, and the bar stays 1e-3."""
    from tennis_amd import weights as W
    enc, frames = setup["enc"], setup["frames"]
    x32 = W.normalize_to_nchw_f32(frames)                      # fp32, un-rounded
    assert (x32.astype(np.float16).astype(np.float32) != x32).any()
    ref = dn.densenet121_features(x32, setup["p"])
    f_u8 = enc(torch.from_numpy(frames).cuda()).cpu().numpy()
    f_nchw = enc(torch.from_numpy(x32).cuda()).cpu().numpy()
    e_u8, e_nchw = float(np.abs(f_u8 - ref).max()), float(np.abs(f_nchw - ref).max())
    report["encoder_u8_vs_unrounded_oracle_maxabs_err"] = e_u8
    report["encoder_nchw_f32_vs_unrounded_oracle_maxabs_err"] = e_nchw
    assert e_u8 < TOL and e_nchw < TOL, (e_u8, e_nchw)


def test_frame_logits(setup, report):
    from tennis_amd.engine import Dense
    p, enc = setup["p"], setup["enc"]
    xd = torch.from_numpy(np.ascontiguousarray(setup["x16"].transpose(0, 2, 3, 1))).cuda()
    cls = Dense(p["framemodel0_dense0_weight"], p["framemodel0_dense0_bias"])
    logits = cls(enc(xd)).cpu().numpy()
    ref = dn.dense(setup["ref"], p, "framemodel0_dense0_")
    e = float(np.abs(logits - ref).max())
    report["frame_logits_maxabs_err"] = e
    assert e < TOL, e


def test_determinism(setup):
    enc = setup["enc"]
    xd = torch.from_numpy(np.ascontiguousarray(setup["x16"].transpose(0, 2, 3, 1))).cuda()
    a = enc(xd).cpu().numpy()
    b = enc(xd).cpu().numpy()
    assert np.array_equal(a, b)


def test_repeatable_bit_for_bit_under_load():
    """Timing-dependent faults (a missing wait state, a race on an LDS ring slot) show up as run-to-run differences
    long before they show up against a tolerance: 256 frames through the split, pipelined encoder thirty times, other
    work queued in between, every run bit-identical to the first - features and the fused stem's output."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    enc = DenseNet121Features(p, 224, max_batch=256)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    x = torch.randn((256, 224, 224, 3), generator=g, device="cuda").half()
    other = torch.randn((4096, 4096), device="cuda")
    ref_f = enc(x).clone()
    ref_p = enc.read_tap("pool0", 4).copy()
    enc.set_pipelined(True)
    out = [torch.empty_like(ref_f) for _ in range(2)]
    for i in range(30):
        if i % 3 == 0: other @ other                 # unrelated kernels on the caller's stream
        enc(x, out=out[i & 1])
        enc.join(0)
        assert torch.equal(out[i & 1], ref_f), f"run {i}: features differ from the first run"
        if i % 10 == 9:
            assert np.array_equal(enc.read_tap("pool0", 4), ref_p), f"run {i}: stem output differs"
    enc.set_pipelined(False)


def test_chained_blocks_match_per_layer_launches(monkeypatch):
    """Default: the 14x14 and 7x7 blocks run all their layers inside one launch per block (one workgroup per
    frame, the next layer's first stages and tables requested during the current layer's store); TN_NO_CHAIN=1
    launches every layer separately.  Same arithmetic in the same order -> bit-identical features.  (TN_NO_BLOCK7: the 7x7
    block on the tile kernel too, as in exact-weights mode, instead of dense_block7.hip.)"""
    import os
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(5)
    x = torch.from_numpy(W.normalize_to_nchw_f32(W.synthetic_frames_u8(8, 224))).cuda()
    monkeypatch.setenv("TN_NO_BLOCK7", "1")
    monkeypatch.setenv("TN_NO_CHAIN", "1")
    ref = DenseNet121Features(p, 224, max_batch=8)(x).cpu().numpy()
    monkeypatch.delenv("TN_NO_CHAIN", raising=False)
    got = DenseNet121Features(p, 224, max_batch=8)(x).cpu().numpy()
    assert np.array_equal(ref, got)


def test_lds_resident_7x7_block_vs_tile_kernel(setup, report, monkeypatch):
    """Default: the 7x7 block runs on dense_block7.hip (concat buffer in LDS, K split over four waves); TN_NO_BLOCK7=1 puts it
    back on the chained tile kernel.  Different summation orders: both within the bar of the fp32 oracle, and within fp16
    rounding noise of each other; the default path gives the same bits for a frame whatever the batch around it."""
    from tennis_amd.engine import DenseNet121Features
    x = torch.from_numpy(setup["x16"].astype(np.float32)).cuda()
    got = DenseNet121Features(setup["p"], 224, max_batch=2)(x)
    monkeypatch.setenv("TN_NO_BLOCK7", "1")
    old = DenseNet121Features(setup["p"], 224, max_batch=2)(x)
    monkeypatch.delenv("TN_NO_BLOCK7")
    e_new = float(np.abs(got.cpu().numpy() - setup["ref"]).max()); e_old = float(np.abs(old.cpu().numpy() - setup["ref"]).max())
    report["features_block7_maxabs_err"] = e_new
    report["features_block7_off_maxabs_err"] = e_old
    assert e_new < TOL and e_old < TOL, (e_new, e_old)
    assert float((got - old).abs().max()) < 2e-3
    big = DenseNet121Features(setup["p"], 224, max_batch=34)(x[torch.arange(34, device="cuda") % 2])
    assert torch.equal(big[:2], got) and torch.equal(big[32:], got)


def test_streamed_28x28_block_route(setup, report, monkeypatch):
    """TN_BLOCK28=1: the 28x28 block as ONE launch of dense_block28.hip (four passes of eight rows per layer, every layer's weights
    streamed once per pass, K up to 480 - the layers the strip kernel cannot hold) instead of seven strip launches + five tile-kernel
    launches.  Measured in round 4 at the same time per step as the default route and more energy: opt-in, but held to
    the same bar, and a frame's features do not depend on the batch around it."""
    from tennis_amd.engine import DenseNet121Features
    x = torch.from_numpy(setup["x16"].astype(np.float32)).cuda()
    dflt = DenseNet121Features(setup["p"], 224, max_batch=2)(x)
    monkeypatch.setenv("TN_BLOCK28", "1")
    got = DenseNet121Features(setup["p"], 224, max_batch=2)(x)
    big = DenseNet121Features(setup["p"], 224, max_batch=34)(x[torch.arange(34, device="cuda") % 2])
    monkeypatch.delenv("TN_BLOCK28")
    e = float(np.abs(got.cpu().numpy() - setup["ref"]).max())
    report["features_block28_maxabs_err"] = e
    assert e < TOL, e
    assert float((got - dflt).abs().max()) < 2e-3 and not torch.equal(got, dflt)      # (a different kernel did run)
    assert torch.equal(big[:2], got) and torch.equal(big[32:], got)


def test_unfused_fallback_path(setup, report, monkeypatch):
    """TN_NO_FUSE=1: the layer-wise kernels (conv1x1 / conv3x3 / stem / maxpool) that serve input sizes the fused
    tiles do not cover; same oracle, same tolerance."""
    from tennis_amd.engine import DenseNet121Features
    monkeypatch.setenv("TN_NO_FUSE", "1")
    enc = DenseNet121Features(setup["p"], 224, max_batch=2)
    got = enc(torch.from_numpy(setup["x16"].astype(np.float32)).cuda()).cpu().numpy()
    err = float(np.abs(got - setup["ref"]).max())
    report["features_unfused_maxabs_err"] = err
    assert err < TOL


def test_encoder_512_features_4096(report):
    """data_shape 512 (reference train.py:259: feature size 4096 = 1024 x 2 x 2, NCHW flatten) — blocks of
    128/64/32/16 pixels run on the layer-wise kernels; checked against the torch-CPU fp32 restatement
    (itself pinned to the numpy oracle in tests/test_cpu_oracle.py)."""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(1, 512)).astype(np.float16)
    enc = DenseNet121Features(p, 512, max_batch=1)
    assert enc.feature_dim == 4096
    got = enc(torch.from_numpy(x16.astype(np.float32)).cuda()).cpu().numpy()
    with torch.no_grad():
        ref = TorchDenseNet121(p)(torch.from_numpy(x16.astype(np.float32))).numpy()
    assert got.shape == ref.shape == (1, 4096)
    err = float(np.abs(got - ref).max())
    report["features_512_maxabs_err"] = err
    assert err < TOL
    # from 13 / 22 frames per launch on, the 128 x 128 / 64 x 64 layers up to K = 320 run on the strip kernels (5 / 3 workgroups
    # per frame): same frame, same bar, and the same bits whatever its position in the batch
    xb = torch.from_numpy(x16.astype(np.float32)).cuda().expand(24, -1, -1, -1).contiguous()
    big = DenseNet121Features(p, 512, max_batch=24)(xb).cpu().numpy()
    err = float(np.abs(big[:1] - ref).max())
    report["features_512_strip_maxabs_err"] = err
    assert err < TOL and np.array_equal(big[23:], big[:1])
    # ... and un-rounded fp32 weights at this size: the calibrated conversion (two other frames for the statistics) keeps the bar
    # against the fp32 oracle on the fp32 weights (the hi + lo mode only exists for 224 x 224)
    from tennis_amd.calibrate import calibrated_fp16_model
    p32 = W.make_densenet121_weights(0, fp16_model=False)
    with torch.no_grad():
        ref32 = TorchDenseNet121(p32)(torch.from_numpy(x16.astype(np.float32))).numpy()
    q = calibrated_fp16_model(p32, torch.from_numpy(W.synthetic_frames_u8(2, 512, seed=77)).cuda(), 512)
    u8 = W.synthetic_frames_u8(1, 512)                       # (the decoded frame itself: the stem's integer operand)
    with torch.no_grad():
        ref32 = TorchDenseNet121(p32)(torch.from_numpy(W.normalize_to_nchw_f32(u8))).numpy()
    got = DenseNet121Features(q, 512, max_batch=1)(torch.from_numpy(u8).cuda()).cpu().numpy()
    err = float(np.abs(got - ref32).max())
    report["features_512_fp32_weights_calibrated_maxabs_err"] = err
    assert err < TOL, err


def test_encoder_448_mixed_kernels(report):
    """448 x 448 input: the maps are 112 / 56 / 28 / 14 pixels, i.e. the sizes the fused kernels tile - one block later than
    at 224, with that block's (larger) channel counts.  Every layer must pick a kernel that takes its K: strip kernel up to
    K = 320 in the 56 / 28 blocks, tile kernel up to its table space (256 / 512 / 1024), layer-wise kernels for the rest and
    for the 112-pixel block; checked against the torch-CPU fp32 restatement.  (Round 3 fix: the fused tile kernel used to be
    chosen by map size alone and refused K = 288 at 56 x 56.)"""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(2, 448)).astype(np.float16)
    with torch.no_grad():
        ref = TorchDenseNet121(p)(torch.from_numpy(x16.astype(np.float32))).numpy()
    xd = torch.from_numpy(x16.astype(np.float32)).cuda()
    got = DenseNet121Features(p, 448, max_batch=2)(xd).cpu().numpy()
    assert got.shape == ref.shape == (2, 4096)
    err = float(np.abs(got - ref).max())
    report["features_448_maxabs_err"] = err
    assert err < TOL
    big = DenseNet121Features(p, 448, max_batch=66)(xd[torch.arange(66, device="cuda") % 2]).cpu().numpy()   # strip kernels (batch >= 64)
    err = float(np.abs(big[:2] - ref).max())
    report["features_448_strip_maxabs_err"] = err
    assert err < TOL and np.array_equal(big[64:], big[:2])


@pytest.mark.parametrize("size", [226, 232, 236])
def test_stem_and_encoder_at_odd_input_sizes(size, report):
    """Input sizes off the 224 grid: 232 has partial tiles in both directions (58 x 58 pooled pixels), 236 is not a
    multiple of 8 wide (element-wise patch staging instead of the vector path), 226 has an odd conv map (113 x 113: the
    last pooled row / column reaches past it).  Fused stem + maxpool output (tap pool0) and the final features against
    the numpy oracle, for the three input layouts."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    frames = W.synthetic_frames_u8(1, size)
    x16 = W.normalize_to_nchw_f32(frames).astype(np.float16)
    taps = {}
    ref = dn.densenet121_features(x16.astype(np.float32), p, taps=taps)
    enc = DenseNet121Features(p, size, max_batch=1)
    inputs = {"nhwc_f16": torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda(),
              "nchw_f32": torch.from_numpy(x16.astype(np.float32)).cuda(),
              "nhwc_u8": torch.from_numpy(frames).cuda()}
    for name, x in inputs.items():
        feat = enc(x).cpu().numpy()
        pool0 = enc.read_tap("pool0", 1).reshape(taps["pool0"].shape)
        e0 = float(np.abs(pool0 - taps["pool0"]).max())
        e = float(np.abs(feat - ref).max())
        report[f"stem_{size}_{name}_pool0_maxabs_err"] = e0
        report[f"features_{size}_{name}_maxabs_err"] = e
        assert e0 < 8e-3, (name, e0)          # fp16 storage of values up to ~8: half an ulp is 4e-3
        assert e < TOL, (name, e)


def test_full_batch_256_matches_small_batches(report, monkeypatch):
    """BASELINE.json configs[1] size (256 frames: two-stream split, one-frame-per-workgroup strip kernels in the 56x56 / 28x28
    blocks, XCD-remapped persistent tiles, chained blocks): frames are independent and the per-frame arithmetic order of a
    kernel does not depend on the batch, so every frame's features must equal, bit for bit, what the same frame gives in a
    smaller batch THROUGH THE SAME KERNELS: 256 vs 128 frames on the default path (strip kernels from 64 frames per launch
    on), 256 vs 4 frames with the strip kernels switched off (TN_NO_STRIP: the 8-wave tile kernels at every batch size).
    The two kernel families accumulate in different orders: they agree to fp16 rounding noise, far inside 1e-3."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(2)
    base = torch.from_numpy(W.normalize_to_nchw_f32(W.synthetic_frames_u8(4, 224))).cuda()
    small = DenseNet121Features(p, 224, max_batch=4)(base)
    idx = torch.arange(256, device="cuda") % 4
    big_in = base[idx].permute(0, 2, 3, 1).contiguous().half()          # NHWC fp16, as bench.py feeds it
    big = DenseNet121Features(p, 224, max_batch=256)(big_in)
    mid = DenseNet121Features(p, 224, max_batch=128)(big_in[:128])      # (two half batches of 64: still the strip kernels)
    assert torch.equal(big[:128], mid)
    assert torch.equal(big[:4], big[252:256])
    ref = DenseNet121Features(p, 224, max_batch=4)(base.permute(0, 2, 3, 1).contiguous().half())
    monkeypatch.setenv("TN_NO_STRIP", "1")
    big_tiles = DenseNet121Features(p, 224, max_batch=256)(big_in)
    monkeypatch.delenv("TN_NO_STRIP")
    assert torch.equal(big_tiles, ref[idx])
    report["full_batch_vs_small_batch_bit_identical"] = True
    d = float((big - big_tiles).abs().max())
    report["strip_vs_tile_kernels_maxabs_diff"] = d
    assert d < 2e-3, d          # two fp16 pipelines with independent rounding, each within 1e-3 of the fp32 oracle (test_strip_path_vs_oracle)
    # and the NHWC fp16 hand-over agrees with the reference NCHW fp32 layout to fp16 input rounding
    assert float((ref - small).abs().max()) < 2e-3


def test_full_batch_256_distinct_frames_vs_oracle(report):
    """BASELINE.json configs[1] at FULL size with 256 DISTINCT frames against the fp32 oracle (oracle/torch_ref.py, 15 s of CPU on the
    test box - cross-checked against oracle/densenet_np.py in tests/test_cpu_oracle.py).  262 144 feature values instead of the 2 048
    - 4 096 of the small tests: the largest single error sits AT the bar (measured 1.03e-3 / 9.95e-4 for weight seeds 0 / 2), the
    99.99th percentile at 6.4 - 7.6e-4, a frame's worst feature at 5.3 - 6.3e-4 in the median; the logits of FrameModel's Dense(11)
    (what north_star's bar names first) stay inside 1e-3 with room.  The test pins that picture: it fails if the tail grows."""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.engine import Dense, DenseNet121Features
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    u8 = W.synthetic_frames_u8(256, 224, seed=77)            # decoded frames as the loader hands them over
    ref = TorchDenseNet121(p)(torch.from_numpy(W.normalize_to_nchw_f32(u8))).numpy()      # ToTensor + Normalize in fp32, un-rounded (evaluate.py:96-97)
    ref_logits = dn.dense(ref, p, "framemodel0_dense0_")
    feat_d = DenseNet121Features(p, 224, max_batch=256)(torch.from_numpy(u8).cuda())
    logits = Dense(p["framemodel0_dense0_weight"], p["framemodel0_dense0_bias"])(feat_d).cpu().numpy()
    e = np.abs(feat_d.cpu().numpy() - ref)
    el = float(np.abs(logits - ref_logits).max())
    report["features_b256_distinct_maxabs_err"] = float(e.max())
    report["features_b256_distinct_p9999_err"] = float(np.quantile(e, 0.9999))
    report["features_b256_distinct_over_bar"] = int((e > TOL).sum())
    report["logits_b256_distinct_maxabs_err"] = el
    assert el < TOL, el
    # the bar itself on every one of the 262 144 values (round 4 allowed eight of them up to 1.25e-3: VERDICT r4 weak 1)
    assert e.max() < TOL, (float(e.max()), int((e > TOL).sum()))
    assert np.median(e.max(1)) < 7.5e-4


def test_fp32_weights_exact_mode(report):
    """north_star: "logits within 1e-3 of the MXNet CPU reference", which evaluates fp32 parameters
    (reference models/vision/definitions.py:27-33).  With UN-rounded fp32 conv weights the default fp16 model carries
    the model-conversion error of rounding 6.9 M weights once (measured and reported, ~3e-3); the exact-weights mode
    (TN_ENC_EXACT_WEIGHTS: hi + lo fp16 pairs in the dense layers and transitions) brings features AND logits under
    1e-3 against the fp32 oracle on those same un-rounded weights."""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.engine import Dense, DenseNet121Features
    p = W.make_densenet121_weights(0, fp16_model=False)
    assert any((v.astype(np.float16).astype(np.float32) != v).any() for k, v in p.items() if k.endswith("_weight"))
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(4, 224)).astype(np.float16)
    ref = TorchDenseNet121(p)(torch.from_numpy(x16.astype(np.float32))).numpy()          # fp32 graph, fp32 weights
    ref2 = dn.densenet121_features(x16[:1].astype(np.float32), p)                        # numpy oracle agrees with it
    assert np.abs(ref2 - ref[:1]).max() < 2e-4
    ref_logits = dn.dense(ref, p, "framemodel0_dense0_")
    xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()
    cls = Dense(p["framemodel0_dense0_weight"], p["framemodel0_dense0_bias"])
    out = {}
    for mode in (False, True):
        enc = DenseNet121Features(p, 224, max_batch=4, exact_weights=mode)
        feat = enc(xd)
        out[mode] = (float(np.abs(feat.cpu().numpy() - ref).max()), float(np.abs(cls(feat).cpu().numpy() - ref_logits).max()))
        del enc
    report["fp32_weights_default_mode_feature_err"], report["fp32_weights_default_mode_logits_err"] = out[False]
    report["fp32_weights_exact_mode_feature_err"], report["fp32_weights_exact_mode_logits_err"] = out[True]
    print("fp32 weights: default (fp16 model) feat/logits err %.2e / %.2e, exact mode %.2e / %.2e" % (out[False] + out[True]))
    assert out[True][0] < 1e-3 and out[True][1] < 1e-3, out
    # batch 64 (two-stream split, persistent tiles, chained blocks) is bit-identical to the batch of 4 in this mode too
    enc4, enc64 = (DenseNet121Features(p, 224, max_batch=b, exact_weights=True) for b in (4, 64))
    idx = torch.arange(64, device="cuda") % 4
    assert torch.equal(enc64(xd[idx].contiguous()), enc4(xd)[idx])


def test_fp32_weights_calibrated_rounding(report):
    """The same bar with ONE fp16 number per weight (tennis_amd.calibrate): eight calibration frames (other frames than the
    test's) go through the layer-wise kernels for the mean activation of every convolution input, each weight is rounded to
    the fp16 neighbour that keeps the mean-weighted rounding error of its output row at zero, and the DEFAULT kernels - small
    batch, and the strip / LDS-resident kernels at 128 frames - evaluate the converted model: features and logits within 1e-3
    of the fp32 oracle on the UN-rounded fp32 weights, where plain rounding is off by ~3e-3."""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    from tennis_amd.calibrate import calibrated_fp16_model
    from tennis_amd.engine import Dense, DenseNet121Features
    p = W.make_densenet121_weights(0, fp16_model=False)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(4, 224)).astype(np.float16)
    ref = TorchDenseNet121(p)(torch.from_numpy(x16.astype(np.float32))).numpy()          # fp32 graph, fp32 weights
    ref_logits = dn.dense(ref, p, "framemodel0_dense0_")
    calib = torch.from_numpy(W.synthetic_frames_u8(8, 224, seed=4321)).cuda()            # NHWC u8, as the frame loader hands them over
    q = calibrated_fp16_model(p, calib)
    changed = sum(int((q[k] != W.as_fp16_model(p)[k]).sum()) for k in q if k.endswith("_weight") and q[k].ndim == 4)
    assert changed > 100000                      # a good part of the 6.9 M roundings went the other way
    xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()
    cls = Dense(p["framemodel0_dense0_weight"], p["framemodel0_dense0_bias"])
    feat = DenseNet121Features(q, 224, max_batch=4)(xd)
    e_f, e_l = float(np.abs(feat.cpu().numpy() - ref).max()), float(np.abs(cls(feat).cpu().numpy() - ref_logits).max())
    big = DenseNet121Features(q, 224, max_batch=128)(xd[torch.arange(128, device="cuda") % 4].contiguous())
    e_b = float(np.abs(big[:4].cpu().numpy() - ref).max())
    report["fp32_weights_calibrated_rounding_feature_err"], report["fp32_weights_calibrated_rounding_logits_err"] = e_f, e_l
    report["fp32_weights_calibrated_rounding_feature_err_b128"] = e_b
    print("fp32 weights, calibrated rounding: features %.2e (batch 128: %.2e), logits %.2e" % (e_f, e_b, e_l))
    assert e_f < 1e-3 and e_l < 1e-3 and e_b < 1e-3, (e_f, e_l, e_b)
    # the same through the model surface: get_model(..., conversion="calibrated") keeps the adopted fp32 weights until calibrate()
    from tennis_amd.model_zoo import get_model
    net = get_model("DenseNet121", pretrained=False, conversion="calibrated").features
    net.set_params({k: v for k, v in p.items() if k.startswith("densenet0_")})
    net.calibrate(calib)
    assert torch.equal(net(xd), feat)


def test_strip_path_vs_oracle(report):
    """The batch sizes the benchmark runs (>= 64 frames per launch) take the strip kernels in the 56x56 / 28x28 blocks: 128
    frames (two distinct ones, tiled) against the fp32 oracle on the same fp16-model parameters, same 1e-3 bar as the
    small-batch path of test_encoder_stages."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    x16 = W.normalize_to_nchw_f32(W.synthetic_frames_u8(2, 224)).astype(np.float16)
    ref = dn.densenet121_features(x16.astype(np.float32), p)
    xd = torch.from_numpy(np.ascontiguousarray(x16.transpose(0, 2, 3, 1))).cuda()
    idx = torch.arange(128, device="cuda") % 2
    feat = DenseNet121Features(p, 224, max_batch=128)(xd[idx].contiguous()).cpu().numpy()
    e = float(np.abs(feat - ref[idx.cpu().numpy()]).max())
    report["features_224_strip_path_b128_maxabs_err"] = e
    assert e < TOL, e


@pytest.mark.parametrize("B", [1, 3, 5, 13, 40, 64, 72, 136])
def test_ragged_batch_sizes(B):
    """Batch sizes that are not multiples of 8 take the un-remapped tile order, 40 / 72 are too small or too ragged for
    the two-stream split, 64 is the smallest split batch: every frame must come out exactly as in a batch of 4.  From 64
    frames per launch on (72 un-split, 136 = two halves of 68) the 56x56 / 28x28 blocks run on the strip kernels: those
    frames must come out exactly as in a batch of 128 (two halves of 64)."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(2)
    base = torch.from_numpy(W.normalize_to_nchw_f32(W.synthetic_frames_u8(4, 224))).cuda().permute(0, 2, 3, 1).contiguous().half()
    strip = B >= 72
    if strip:
        ref = DenseNet121Features(p, 224, max_batch=128)(base[torch.arange(128, device="cuda") % 4].contiguous())[:4]
    else:
        ref = DenseNet121Features(p, 224, max_batch=4)(base)
    idx = (torch.arange(B, device="cuda") * 3) % 4
    got = DenseNet121Features(p, 224, max_batch=B)(base[idx].contiguous())
    assert got.shape == (B, 1024) and torch.equal(got, ref[idx])


def test_batch_larger_than_handle_is_an_error():
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(2)
    enc = DenseNet121Features(p, 224, max_batch=2)
    x = torch.zeros((3, 224, 224, 3), dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="max_batch"):
        enc(x)


def test_pipelined_forwards_with_changing_batch_size():
    """Pipelined split forwards share the workspace by row range per side stream; a call with a different batch size (a
    corpus' ragged last batch: 128 behind 256) moves the ranges across the streams and must be ordered behind everything
    the earlier calls left on either of them (api.hip split path)."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    enc = DenseNet121Features(p, 224, max_batch=256)
    big = [torch.from_numpy(W.synthetic_frames_u8(256, 224, seed=s)).cuda() for s in (11, 12)]
    small = torch.from_numpy(W.synthetic_frames_u8(128, 224, seed=13)).cuda()
    ref = [enc(x).clone() for x in (big[0], small, big[1])]
    torch.cuda.synchronize()
    enc.set_pipelined(True)
    for _ in range(3):
        outs = [torch.empty_like(r) for r in ref]
        for x, o in zip((big[0], small, big[1]), outs):
            enc(x, out=o)
        enc.join(0)
        enc.join(1)
        torch.cuda.synchronize()
        for a, b in zip(ref, outs):
            assert torch.equal(a, b)
    enc.set_pipelined(False)


def test_pipelined_forwards_match_joined_ones():
    """tn_densenet121_set_pipelined: consecutive forwards overlap on the side streams and the caller joins them - the
    features are the ones the stream-ordered forwards give, for results consumed at once (lag 0) and one call behind"""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    enc = DenseNet121Features(p, 224, max_batch=64)
    xs = [torch.from_numpy(W.synthetic_frames_u8(64, 224, seed=s)).cuda() for s in range(4)]
    ref = [enc(x).clone() for x in xs]
    torch.cuda.synchronize()
    enc.set_pipelined(True)
    outs = [torch.empty_like(ref[0]) for _ in xs]
    got = []
    for i, x in enumerate(xs):
        enc(x, out=outs[i])
        if i > 0:
            enc.join(1)
            got.append(outs[i - 1].clone())         # on torch's stream, behind the join of call i-1
    enc.join(0)
    got.append(outs[-1].clone())
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    small = enc(xs[0][:4])                           # an un-split call in the mode: ordered behind everything outstanding
    enc.set_pipelined(False)
    assert torch.equal(small, ref[0][:4]) or float((small - ref[0][:4]).abs().max()) < 1e-3
    again = enc(xs[1])
    torch.cuda.synchronize()
    assert torch.equal(again, ref[1])


def test_a_nan_batch_does_not_poison_the_next_one():
    """the streamed 14x14 block reads the 32 channels a layer is about to write as the zero-weighted pad of its last
    super-step, and those hold whatever the previous forward of the frame slot left there.  With BN1 as a clamp the pad's
    constants are lo = hi = 0 and v_pk_max_f16(NaN, 0) = 0: a forward whose input was NaN / inf (a corrupt frame handed over as
    fp32) leaves nothing behind - the next batch through the same encoder gives the bits a fresh encoder gives."""
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features
    p = W.make_densenet121_weights(0)
    good = torch.from_numpy(W.synthetic_frames_u8(4, 224, seed=21)).cuda()
    want = DenseNet121Features(p, 224, max_batch=4)(good).cpu().numpy()
    enc = DenseNet121Features(p, 224, max_batch=4)
    bad = torch.full((4, 3, 224, 224), float("nan"), dtype=torch.float32, device="cuda")
    bad[1] = float("inf")
    bad[2] = -60000.0
    enc(bad)               # (v_pk_max_f16 / v_max_f32 return the non-NaN operand: even these frames' features come out finite)
    got = enc(good).cpu().numpy()
    assert np.isfinite(got).all() and np.array_equal(got, want)
