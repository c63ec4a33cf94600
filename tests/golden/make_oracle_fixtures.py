"""Regression fixtures of the CPU ORACLE (oracle/) on seeded inputs — the list of SURVEY.md §8(c):
densenet121_224_b2, bigru / bilstm (b2 t8 f1024 h128), gnmt_step, beam_trace.

These pin the oracle to itself (they detect drift of the restatement and give the GPU tests committed numbers to
compare with); they are NOT reference outputs — MXNet / GluonCV / GluonNLP cannot be installed, so the oracle stays
"parity unpinned" for those ops.  Inputs are regenerated from the seeds recorded in each file.

Run:  python tests/golden/make_oracle_fixtures.py        (writes tests/golden/oracle_*.npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import densenet_np as dn          # noqa: E402
from oracle import gnmt_np as gn              # noqa: E402
from oracle import rnn_np as rn               # noqa: E402
from tennis_amd import weights as W           # noqa: E402


def densenet_inputs():
    p = W.make_densenet121_weights(0)
    p.update(W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_"))
    x = W.normalize_to_nchw_f32(W.synthetic_frames_u8(2, 224, 1234))      # ToTensor + Normalize in fp32, un-rounded: what the reference's network sees (evaluate.py:96-97)
    return p, x


def make_densenet():
    p, x = densenet_inputs()
    taps = {}
    feats = dn.densenet121_features(x, p, taps=taps)      # fp32 weights: the 'fp32 oracle' of the parity bar
    logits = dn.dense(feats, p, "framemodel0_dense0_")
    out = {"feats": feats.astype(np.float32), "logits": logits.astype(np.float32)}
    for k, v in taps.items():
        out[f"tap_{k}_mean"] = np.float32(v.mean())
        out[f"tap_{k}_absmax"] = np.float32(np.abs(v).max())
    np.savez(os.path.join(HERE, "oracle_densenet121_224_b2.npz"), weights_seed=0, dense_seed=1, frames_seed=1234, **out)


def rnn_inputs(mode):
    p = W.make_rnn_weights(3, mode, 1024, 128, "rnn_")
    x = (np.abs(np.random.default_rng(7).normal(0, 1, (2, 8, 1024))) * 0.5).astype(np.float32)
    return p, x, np.array([8, 5], np.int32)


def make_rnn(mode):
    p, x, vl = rnn_inputs(mode)
    full, (fh, _), (bh, _) = rn.birnn_layer(x, p, "rnn_", mode, None)
    ragged, (rfh, _), (rbh, _) = rn.birnn_layer(x, p, "rnn_", mode, vl)
    np.savez(os.path.join(HERE, f"oracle_bi{mode}_b2_t8_f1024.npz"), weights_seed=3, input_seed=7, valid_length=vl,
             seq=full.astype(np.float32), h_fwd=fh.astype(np.float32), h_bwd=bh.astype(np.float32),
             seq_ragged=ragged.astype(np.float32), h_fwd_ragged=rfh.astype(np.float32), h_bwd_ragged=rbh.astype(np.float32))


GN = dict(seed=3, B=3, T=19, F=64, H=32, E=20, V=40, beam=4, max_length=24, proj_scale=40.0)


def gnmt_inputs():
    c = GN
    p = W.make_gnmt_weights(c["seed"], "gru", c["F"], c["H"], c["E"], c["V"])
    p["gnmt_tgt_proj_weight"] = (p["gnmt_tgt_proj_weight"] * c["proj_scale"]).astype(np.float32)
    rng = np.random.default_rng(c["seed"])
    src = (np.abs(rng.normal(0, 1, (c["B"], c["T"], c["F"]))) * 0.5).astype(np.float32)
    vl = rng.integers(c["T"] // 3, c["T"] + 1, c["B"]).astype(np.int32)
    vl[0] = c["T"]
    return p, src, vl


def make_gnmt():
    c = GN
    p, src, vl = gnmt_inputs()
    mem, states = gn.encoder(src, vl, p, "gru", c["H"])
    dec = gn.Decoder(p, c["H"], cell="gru")
    rnn_states, att = dec.init_state(mem, states, vl)
    tok = np.array([2, 5, 7], np.int64)
    logp, new_states, ctx = dec.step(tok, rnn_states, att, np.arange(c["B"]))
    np.savez(os.path.join(HERE, "oracle_gnmt_step.npz"), mem=mem.astype(np.float32), tokens=tok, logp=logp.astype(np.float32),
             h0=new_states[0].astype(np.float32), h1=new_states[1].astype(np.float32), ctx=ctx.astype(np.float32), valid_length=vl)
    samples, scores, vlen = gn.beam_search(dec, mem, states, vl, 2, 3, c["beam"], 1.0, 5, c["max_length"])
    np.savez(os.path.join(HERE, "oracle_beam_trace.npz"), samples=samples, scores=scores.astype(np.float32), valid_length=vlen,
             bos=2, eos=3, alpha=1.0, K=5.0, **{f"cfg_{k}": v for k, v in c.items()})


if __name__ == "__main__":
    make_densenet()
    make_rnn("gru")
    make_rnn("lstm")
    make_gnmt()
    for f in sorted(os.listdir(HERE)):
        if f.startswith("oracle_"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
