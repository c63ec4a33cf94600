"""Secondary stages of the path at the BASELINE.json configs (not the headline metric):
  C3  CNN features -> bi-GRU(128) -> max over T -> Dense(11): clip batch 32 x T=64 x F=1024
  C5  GNMT captioner on features: B=32 clips, T=214, F=1024, H=256, E=100, V=254, beam 5, max_len 150
  input side: Resize(256)+CenterCrop(224) of 720p uint8 frames (tn_preproc_*)
Prints one JSON object; inputs are synthetic and resident on the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import weights as W
from tennis_amd.engine import BiRNN, Dense, GNMTCaptioner, TemporalHeadTrainer, temporal_pool

dev = torch.device("cuda:0")


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


out = {}
rng = np.random.default_rng(0)
# ---- C3 -------------------------------------------------------------------------------------------------------
B, T, F, H = 32, 64, 1024, 128
p = W.make_rnn_weights(0, "gru", F, H, "cnnrnn0_gru0_")
p.update(W.make_dense_weights(1, 11, 2 * H, "cnnrnn0_dense0_"))
rnn = BiRNN("gru", F, H, p, "cnnrnn0_gru0_", max_rows=B * T)
fc = Dense(p["cnnrnn0_dense0_weight"], p["cnnrnn0_dense0_bias"])
x = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
def c3():
    seq = rnn(x)
    return fc(temporal_pool(seq, "max"))
s = timed(c3, 50)
out["C3_bigru_head"] = {"ms_per_clip_batch": round(s * 1e3, 3), "clips_per_s": round(B / s, 1), "frames_per_s": round(B * T / s, 1),
                        "gflop": 3.624, "tflops": round(3.624e9 / s / 1e12, 3)}
tr = TemporalHeadTrainer(p, F, H, 11, max_batch=B, max_steps=T)
yl = torch.from_numpy(rng.integers(0, 11, B).astype(np.int32)).to(dev)
def c3t():
    tr.forward_backward(x, yl)
    tr.step(B, 1e-3, 0.9, 1e-4)
s = timed(c3t, 50)
out["C3_train_step"] = {"ms_per_clip_batch": round(s * 1e3, 3), "clips_per_s": round(B / s, 1),
                        "note": "forward + softmax CE + BPTT + weight-gradient GEMMs + SGD momentum update, fp32"}
# ---- C3 end to end, IMAGE mode: 32 clips x 64 frames = 2 048 decoded frames -> DenseNet-121 encoder ->
# bi-GRU(128) -> max over T -> Dense(11), the whole CNNRNN of reference models/vision/definitions.py:75-110 on frames -----------
if os.environ.get("TN_STAGES_SKIP_C3_IMAGE") is None:
    from tennis_amd.calibrate import calibrated_fp16_model
    from tennis_amd.engine import DenseNet121Features
    B3, T3 = 32, 64
    pe = calibrated_fp16_model(W.make_densenet121_weights(0, fp16_model=False), None, 224)
    enc = DenseNet121Features(pe, 224, max_batch=256)
    g = torch.Generator(device=dev); g.manual_seed(3)
    frames = torch.randint(0, 256, (B3 * T3, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
    feats = torch.empty((B3 * T3, 1024), dtype=torch.float32, device=dev)
    enc.set_pipelined(True)
    def c3_image():
        for i in range(0, B3 * T3, 256):
            enc(frames[i:i + 256], out=feats[i:i + 256])
        enc.join(0); enc.join(1)
        seq = rnn(feats.view(B3, T3, 1024))
        return fc(temporal_pool(seq, "max"))
    s = timed(c3_image, 10)
    def c3_enc_only():
        for i in range(0, B3 * T3, 256):
            enc(frames[i:i + 256], out=feats[i:i + 256])
        enc.join(0); enc.join(1)
    s_e = timed(c3_enc_only, 10)
    enc.set_pipelined(False)
    out["C3_image_mode_end_to_end"] = {"clips": B3, "frames_per_clip": T3, "ms_per_clip_batch": round(s * 1e3, 3), "clips_per_s": round(B3 / s, 1),
                                       "frames_per_s": round(B3 * T3 / s, 1), "encoder_only_ms": round(s_e * 1e3, 3),
                                       "temporal_head_ms": round((s - s_e) * 1e3, 3),
                                       "note": "2 048 uint8 224x224 frames resident in HBM -> encoder in 8 pipelined batches of 256 -> bi-GRU -> max -> Dense(11); logits (32, 11)"}
    del enc, frames
# ---- C5 -------------------------------------------------------------------------------------------------------
B, T, F, H, E, V, beam, ml = 32, 214, 1024, 256, 100, 254, 5, 150
p = W.make_gnmt_weights(0, "gru", F, H, E, V)
cap = GNMTCaptioner(p, F, H, E, V, beam=beam, max_length=ml, max_batch=B, max_src_len=T)
src = torch.from_numpy(np.abs(rng.normal(0, 1, (B, T, F))).astype(np.float32) * 0.5).to(dev)
vl = torch.from_numpy(np.clip(rng.integers(60, 600, B), 1, T).astype(np.int32)).to(dev)
s_enc = timed(lambda: cap.encode(src, vl), 20)
def c5():
    cap.encode(src, vl)
    return cap.beam_search(2, 3, 1.0, 5.0)
s_all = timed(c5, 5)
smp, sc, svl = c5()
out["C5_gnmt"] = {"encode_ms": round(s_enc * 1e3, 2), "encode_plus_beam_ms": round(s_all * 1e3, 2),
                  "clips_per_s": round(B / s_all, 1), "steps_run": int(svl.max().item()) - 1,
                  "note": "random-init weights: beams rarely emit EOS, so this is the max_length=150 worst case"}
# ---- C5 training step: 32 clips x T=214 x F=1024, captions of 20 tokens ---------------------------------------------
from tennis_amd.engine import GNMTTrainer
gtr = GNMTTrainer(p, F, H, E, V, max_batch=B, max_src_len=T, max_tgt_len=20)
tg = torch.from_numpy(rng.integers(4, V, (B, 20)).astype(np.int32)).to(dev)
tg[:, 0] = 2
tv = torch.full((B,), 20, dtype=torch.int32, device=dev)
def c5t():
    gtr.forward_backward(src, vl, tg, tv)
    gtr.step(1e-3)
s = timed(c5t, 10)
out["C5_train_step"] = {"ms_per_clip_batch": round(s * 1e3, 2), "clips_per_s": round(B / s, 1),
                        "note": "teacher-forced forward (19 decoder steps) + backward through decoder, attention and both encoder layers + Adam, fp32"}
# ---- end-to-end fine-tuning step of the frame classifier (fp32, training-mode BatchNorm) ---------------------------
from tennis_amd.engine import FrameModelTrainer
pf = W.make_densenet121_weights(0)
pf.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
BF = 16
ftr = FrameModelTrainer(pf, 224, 11, batch=BF)
xf = torch.randn((BF, 224, 224, 3), device=dev)
yf = torch.randint(0, 11, (BF,), dtype=torch.int32, device=dev)
def ftstep():
    ftr.forward_backward(xf, yf)
    ftr.step(BF, 1e-3, 0.9, 1e-4)
s = timed(ftstep, 5)
out["finetune_step_b16"] = {"ms_per_batch": round(s * 1e3, 1), "frames_per_s": round(BF / s, 1), "tflops": round(3 * 5.666e9 * BF / s / 1e12, 2),
                            "note": "correct-first path: every convolution an exact-f32 MFMA GEMM (3x3 via im2col), 3 x forward FLOPs counted"}
del ftr
# ---- input side: Resize(256) + CenterCrop(224) of 256 decoded 720p frames, resident in HBM ---------------------------
from tennis_amd import transforms as TT
tf = TT.Compose([TT.Resize(256), TT.CenterCrop(224), TT.ToTensor(), TT.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
fr = torch.randint(0, 256, (256, 720, 1280, 3), dtype=torch.uint8, device=dev)
s = timed(lambda: tf(fr), 20)
out["resize_crop_720p"] = {"ms_per_256_frames": round(s * 1e3, 3), "frames_per_s": round(256 / s, 1),
                           "note": "4 source pixels per output pixel: 0.6 MB of the 2.76 MB frame are touched"}
print(json.dumps(out))
