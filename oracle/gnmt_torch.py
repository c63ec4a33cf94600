"""ORACLE CROSS-CHECK (test infrastructure only) — the captioner's decode step and beam search written a second
time, on torch's own building blocks (``nn.GRUCell`` / ``nn.LSTMCell``, ``F.softmax`` / ``F.log_softmax``,
``torch.topk``), to give oracle/gnmt_np.py an independent check: the two restatements share no arithmetic code.

What is restated (same reference lines as gnmt_np.py):
  GNMTDecoder.hybrid_forward (one step)   reference models/captioning/gnmt.py:369-404
  NMTModel.decode_step + log_softmax      reference utils/translation.py:51-53
  BeamSearchSampler / BeamSearchScorer    [EXT gluonnlp], driven as reference utils/translation.py:66-82

It does NOT pin the oracle to gluonnlp (absent here: PARITY UNPINNED) — it removes the "one
restatement checked against itself" weakness.  torch's cells use the reference's gate orders ([r, z, n] and
[i, f, g, o]), so the Gluon parameter arrays load as they are.
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _cell(kind, p, name, in_size, hidden):
    c = (torch.nn.GRUCell if kind == "gru" else torch.nn.LSTMCell)(in_size, hidden).double()
    with torch.no_grad():
        c.weight_ih.copy_(torch.from_numpy(p[name + "i2h_weight"]).double())
        c.weight_hh.copy_(torch.from_numpy(p[name + "h2h_weight"]).double())
        c.bias_ih.copy_(torch.from_numpy(p[name + "i2h_bias"]).double())
        c.bias_hh.copy_(torch.from_numpy(p[name + "h2h_bias"]).double())
    return c


class TorchDecoder:
    """One-step decoder of the 2-cell GNMT stack (reference flag defaults) in float64."""

    def __init__(self, p, hidden, embed, cell="gru", prefix="gnmt_"):
        self.kind, self.h = cell, hidden
        t = lambda k: torch.from_numpy(np.asarray(p[prefix + k])).double()
        self.embed = t("tgt_embed_weight")
        self.key_w = t("dec_attention_key_weight")
        self.proj_w, self.proj_b = t("tgt_proj_weight"), t("tgt_proj_bias")
        self.c0 = _cell(cell, p, prefix + "dec_rnn0_", embed + hidden, hidden)
        self.c1 = _cell(cell, p, prefix + "dec_rnn1_", 2 * hidden, hidden)

    def init(self, mem, valid_length):
        self.mem = torch.from_numpy(mem).double()
        self.key = self.mem @ self.key_w.T                       # Dense(H, no bias) on the key only (luong style)
        t = self.mem.shape[1]
        self.mask = torch.arange(t)[None, :] < torch.as_tensor(np.asarray(valid_length))[:, None]

    @torch.no_grad()
    def step(self, tokens, states, att, rows):
        """states: GRU [h0, h1]; LSTM [h0, c0, h1, c1] (tensors (R,H)).  -> (log-probabilities (R,V), states, context)"""
        x = torch.cat([self.embed[torch.as_tensor(tokens)], att], dim=1)
        if self.kind == "gru":
            h0 = self.c0(x, states[0])
            new = [h0]
        else:
            h0, c0 = self.c0(x, (states[0], states[1]))
            new = [h0, c0]
        score = torch.einsum("rh,rth->rt", h0 / np.sqrt(self.h), self.key[rows])      # scaled dot product
        m = self.mask[rows]
        w = F.softmax(score.masked_fill(~m, -1e18), dim=-1) * m
        ctx = torch.einsum("rt,rth->rh", w, self.mem[rows])
        x1 = torch.cat([h0, ctx], dim=1)
        if self.kind == "gru":
            h1 = self.c1(x1, states[1])
            new.append(h1)
        else:
            h1, c1 = self.c1(x1, (states[2], states[3]))
            new += [h1, c1]
        return F.log_softmax(h1 @ self.proj_w.T + self.proj_b, dim=-1), new, ctx


@torch.no_grad()
def beam_search(dec: TorchDecoder, mem, enc_states, valid_length, bos, eos, beam, alpha, K, steps):
    """``steps`` steps of BeamSearchSampler with BeamSearchScorer(alpha, K), candidates ranked by ``torch.topk``
    (no early exit, no final EOS append: the comparison is step by step).  enc_states as gnmt_np.encoder returns them.
    -> samples (B, beam, 1 + steps) int64, scores (B, beam), alive (B, beam)"""
    B, H = mem.shape[0], dec.h
    dec.init(mem, valid_length)
    tt = lambda a: torch.from_numpy(np.asarray(a)).double()
    if dec.kind == "lstm":
        states = [tt(a) for s in enc_states for a in (s[0], s[1])]
    else:
        states = [tt(s[0]) for s in enc_states]
    rows = torch.arange(B).repeat_interleave(beam)
    states = [s.repeat_interleave(beam, dim=0) for s in states]
    att = torch.zeros((B * beam, mem.shape[2]), dtype=torch.float64)
    tokens = torch.full((B * beam,), bos, dtype=torch.long)
    scores = torch.zeros((B, beam), dtype=torch.float64)
    scores[:, 1:] = -1e18
    alive = torch.ones((B, beam), dtype=torch.bool)
    samples = torch.full((B, beam, 1), bos, dtype=torch.long)
    lp = lambda n: ((K + n) / (K + 1.0)) ** alpha
    for i in range(steps):
        step = i + 1
        logp, new_states, new_att = dec.step(tokens, states, att, rows)
        V = logp.shape[1]
        prev = 1.0 if step == 1 else lp(step - 1)
        cand = ((scores * prev)[:, :, None] + logp.view(B, beam, V)) / lp(step)
        cand = torch.where(alive[:, :, None], cand, torch.tensor(-1e18, dtype=torch.float64))
        fin = torch.where(alive, torch.tensor(-1e18, dtype=torch.float64), scores)
        allc = torch.cat([cand.view(B, -1), fin], dim=1)
        new_scores, idx = torch.topk(allc, beam, dim=1)
        use_prev = idx >= beam * V
        word = torch.where(use_prev, torch.tensor(-1), idx % V)
        beam_id = torch.where(use_prev, idx - beam * V, idx // V)
        flat = (beam_id + torch.arange(B)[:, None] * beam).view(-1)
        samples = torch.cat([samples.view(B * beam, -1)[flat].view(B, beam, -1), word[:, :, None]], dim=2)
        states = [s[flat] for s in new_states]
        att = new_att[flat]
        alive = alive.view(-1)[flat].view(B, beam) & (word != eos)
        scores = new_scores
        tokens = word.clamp(min=0).view(-1)
    return samples.numpy(), scores.numpy(), alive.numpy()
