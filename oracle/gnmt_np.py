"""ORACLE (test infrastructure only) — fp32 CPU restatement of the captioner.

In-repo logic restated line by line:
  GNMTEncoder.forward                 reference models/captioning/gnmt.py:136-160
  GNMTDecoder.init_state_from_encoder reference models/captioning/gnmt.py:224-252
  GNMTDecoder.hybrid_forward (step)   reference models/captioning/gnmt.py:345-404
  BeamSearchTranslator.translate      reference utils/translation.py:51-82
  captioning evaluate (ids -> tokens) reference train_gnmt.py:287-300

PARITY UNPINNED for the third-party pieces (gluonnlp 0.x, absent here):
  * NMTModel wiring: decode_step = tgt_proj(decoder(tgt_embed(tok), states));
  * 'scaled_luong' attention = dot-product attention, query / sqrt(H), bias-free Dense(H)
    on the KEY only, masked softmax (masked scores -> -1e18, weights re-multiplied by mask);
  * BeamSearchScorer(alpha, K): LP(n) = ((K+n)/(K+1))^alpha,
    candidate = (score * LP(step-1 | 1 at step 1) + logp) / LP(step);
  * BeamSearchSampler: beams 1.. start at -1e18; candidates = [beam*V scores | finished
    scores]; top-`beam` (descending); finished beams keep their score and emit -1; states
    are re-gathered by beam id; stops when every beam is finished, else after max_length
    steps appends EOS to the unfinished ones.  Returns int32 samples with leading BOS.
They follow the published gluonnlp sources as recalled in the survey; the reference repo holds
no test or golden vector for them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

from . import rnn_np as rn

NEG = np.float32(-1e18)


def encoder(x, valid_length, p, cell="gru", hidden=128, num_layers=2, num_bi_layers=1, prefix="gnmt_enc_", use_residual=False):
    """gnmt.py:136-160 (dropout off).  x (B,T,F) -> (mem (B,T,H), states)."""
    states = []
    inp = x
    for i in range(num_layers):
        if i < num_bi_layers:
            q = {k.replace(f"{prefix}rnn{i}_l_", "L_l0_").replace(f"{prefix}rnn{i}_r_", "L_r0_"): v
                 for k, v in p.items() if k.startswith(f"{prefix}rnn{i}_")}
            out, (fh, fc), (bh, bc) = rn.birnn_layer(inp, q, "L_", cell, valid_length)
            states.append((bh, bc))                 # gnmt.py:146-148: the BACKWARD cell's final state
        else:
            out, h, c = rn.rnn_direction(inp, p, f"{prefix}rnn{i}_", cell, False, valid_length)
            states.append((h, c))
        if use_residual and i > num_bi_layers:      # gnmt.py:155-157
            out = (out + inp).astype(np.float32)
        inp = out
    # SequenceMask (gnmt.py:157-159): padded steps are already zero in rnn_direction's output
    return inp, states


def _sig(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def _log_softmax(z):
    m = z.max(axis=-1, keepdims=True)
    e = np.exp(z - m)
    return (z - m - np.log(e.sum(axis=-1, keepdims=True))).astype(np.float32)


class Decoder:
    """One-step GNMT decoder + target embedding + projection (GRU or LSTM cells).  The recurrent state travels
    as a flat list of (R,H) arrays: [h0, h1] for GRU, [h0, c0, h1, c1] for LSTM (the cell output is h)."""

    def __init__(self, p, hidden, num_layers=2, prefix="gnmt_", cell="gru", use_residual=False):
        self.p, self.h, self.nl, self.pre, self.cell, self.residual = p, hidden, num_layers, prefix, cell, use_residual

    def init_state(self, mem, enc_states, valid_length):
        """gnmt.py:224-252"""
        b, t, _ = mem.shape
        self.mem = mem
        self.keyproj = (mem @ self.p[self.pre + "dec_attention_key_weight"].T).astype(np.float32)
        self.mask = (np.arange(t)[None, :] < np.asarray(valid_length)[:, None])
        if self.cell == "lstm":                       # gnmt.py:224-252: the encoder's [h, c] per layer
            return [a.copy() for s in enc_states for a in (s[0], s[1])], np.zeros((b, mem.shape[2]), np.float32)
        return [s[0].copy() for s in enc_states], np.zeros((b, mem.shape[2]), np.float32)

    def step(self, tokens, rnn_states, att, rows):
        """gnmt.py:369-404 + NMTModel.decode_step + log_softmax (translation.py:51-53).
        `rows` maps each decoder row to its source clip (mem / mask row)."""
        p, pre, H = self.p, self.pre + "dec_", self.h
        emb = p[self.pre + "tgt_embed_weight"][tokens]
        x = np.concatenate([emb, att], axis=-1)
        lstm = self.cell == "lstm"

        def cell(i, inp):
            w = [p[f"{pre}rnn{i}_i2h_weight"], p[f"{pre}rnn{i}_h2h_weight"], p[f"{pre}rnn{i}_i2h_bias"], p[f"{pre}rnn{i}_h2h_bias"]]
            if lstm:
                return rn.lstm_cell(inp, rnn_states[2 * i], rnn_states[2 * i + 1], *w)
            return rn.gru_cell(inp, rnn_states[i], *w), None

        new_states = []
        h0, c0 = cell(0, x)
        new_states += [h0, c0] if lstm else [h0]
        q = (h0 / np.float32(np.sqrt(H))).astype(np.float32)
        score = np.einsum("rh,rth->rt", q, self.keyproj[rows]).astype(np.float32)
        m = self.mask[rows]
        score = np.where(m, score, NEG)
        e = np.exp(score - score.max(axis=-1, keepdims=True))
        w = (e / e.sum(axis=-1, keepdims=True)).astype(np.float32) * m
        ctx = np.einsum("rt,rth->rh", w, self.mem[rows]).astype(np.float32)
        out = h0
        for i in range(1, self.nl):
            cur = out
            hi, ci = cell(i, np.concatenate([cur, ctx], axis=-1))
            new_states += [hi, ci] if lstm else [hi]
            out = (hi + cur).astype(np.float32) if self.residual else hi      # gnmt.py:394-395: the STATE stays the cell's
        logits = out @ p[self.pre + "tgt_proj_weight"].T + p[self.pre + "tgt_proj_bias"]
        self.last_logits = logits.astype(np.float32)        # the un-normalised projection of this step (decode_seq)
        return _log_softmax(logits.astype(np.float32)), new_states, ctx


def beam_search(dec: Decoder, mem, enc_states, valid_length, bos, eos, beam=4, alpha=1.0, K=5, max_length=150):
    """gluonnlp BeamSearchSampler + BeamSearchScorer [EXT]; see module docstring."""
    B = mem.shape[0]
    rnn_states, att = dec.init_state(mem, enc_states, valid_length)
    rows = np.repeat(np.arange(B), beam)
    rnn_states = [np.repeat(s, beam, axis=0) for s in rnn_states]
    att = np.repeat(att, beam, axis=0)
    step_input = np.full(B * beam, bos, np.int64)
    vlen = np.ones((B, beam), np.int32)
    scores = np.zeros((B, beam), np.float32)
    scores[:, 1:] = NEG
    alive = np.ones((B, beam), bool)
    samples = np.full((B, beam, 1), bos, np.int32)
    lp = lambda n: np.float32((K + n) ** alpha / (K + 1) ** alpha)
    finished_early = False
    for i in range(max_length):
        step = i + 1
        logp, new_states, new_att = dec.step(step_input, rnn_states, att, rows)
        V = logp.shape[1]
        prev_lp = np.float32(1.0) if step == 1 else lp(step - 1)
        cand = ((scores * prev_lp)[:, :, None] + logp.reshape(B, beam, V)) / lp(step)
        cand = np.where(alive[:, :, None], cand, NEG).astype(np.float32)
        fin = np.where(alive, NEG, scores).astype(np.float32)
        allc = np.concatenate([cand.reshape(B, -1), fin], axis=1)
        idx = np.argsort(-allc, axis=1, kind="stable")[:, :beam]
        new_scores = np.take_along_axis(allc, idx, axis=1)
        use_prev = idx >= beam * V
        word = np.where(use_prev, -1, idx % V).astype(np.int32)
        beam_id = np.where(use_prev, idx - beam * V, idx // V)
        flat = (beam_id + np.arange(B)[:, None] * beam).reshape(-1)
        samples = np.concatenate([samples.reshape(B * beam, -1)[flat].reshape(B, beam, -1), word[:, :, None]], axis=2)
        vlen = vlen.reshape(-1)[flat].reshape(B, beam) + 1 - use_prev.astype(np.int32)
        rnn_states = [s[flat] for s in new_states]
        att = new_att[flat]
        alive = alive.reshape(-1)[flat].reshape(B, beam) & (word != eos)
        scores = new_scores
        step_input = np.maximum(word, 0).reshape(-1).astype(np.int64)
        if not alive.any():
            finished_early = True
            break
    if not finished_early:
        final = np.where(alive, eos, -1).astype(np.int32)
        samples = np.concatenate([samples, final[:, :, None]], axis=2)
        vlen = vlen + alive.astype(np.int32)
    return samples.astype(np.int32), scores.astype(np.float32), vlen.astype(np.int32)


def ids_to_sentences(samples, vlen, idx_to_token):
    """train_gnmt.py:289-294: best beam, strip BOS/EOS via [1 : valid_len-1]."""
    return [[idx_to_token[int(t)] for t in samples[i, 0, 1:int(vlen[i, 0]) - 1]] for i in range(samples.shape[0])]


def decode_seq(dec: Decoder, mem, enc_states, valid_length, tgt):
    """Teacher forcing: NMTModel.forward -> GNMTDecoder.decode_seq (gnmt.py:254-304) as called at
    train_gnmt.py:280.  tgt (B,L) ids -> logits (B,L,V) (log-softmax NOT applied)."""
    B, L = tgt.shape
    rnn_states, att = dec.init_state(mem, enc_states, valid_length)
    rows = np.arange(B)
    p, pre = dec.p, dec.pre
    outs = []
    for i in range(L):
        logp, rnn_states, att = dec.step(np.maximum(tgt[:, i], 0), rnn_states, att, rows)
        outs.append(dec.last_logits)
    return np.stack(outs, axis=1)


def masked_softmax_ce(logits, labels, valid_length):
    """gluonnlp MaskedSoftmaxCELoss [EXT]: mean over the L steps of -logp[label] * (t < valid_len)."""
    B, L, V = logits.shape
    logp = _log_softmax(logits)
    nll = -np.take_along_axis(logp, labels[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    mask = (np.arange(L)[None, :] < np.asarray(valid_length)[:, None])
    return (nll * mask).mean(axis=1).astype(np.float32)
