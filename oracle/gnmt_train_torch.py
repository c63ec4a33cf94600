"""ORACLE (test infrastructure only) — the captioner's TRAINING step on the CPU with torch autograd (float64).

The forward pass is oracle/gnmt_np.py's (same equations, same conventions; checked against it in
tests/test_cpu_oracle.py::test_gnmt_train_oracle_forward_matches_numpy_oracle) written with torch ops so that autograd
gives the gradients:
  model    NMTModel(src, tgt[:, :-1], src_valid_length, tgt_valid_length - 1)          reference train_gnmt.py:331
           = GNMTEncoder.forward (gnmt.py:136-160) -> GNMTDecoder.decode_seq (gnmt.py:254-304) -> tgt_proj
  loss     MaskedSoftmaxCELoss(out, tgt[:, 1:], tgt_valid_length - 1).mean()
           * (tgt.shape[1] - 1) / (tgt_valid_length - 1).mean()                         train_gnmt.py:332-333
           = summed negative log-likelihood of the valid target tokens / their number
  update   gluon.Trainer(params, 'adam', {'learning_rate': lr}).step(1)                 train_gnmt.py:310,337
           MXNet Adam [EXT]: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); m, v moments; w -= lr_t * m / (sqrt(v) + eps)

GRU (the reference's flag default) and LSTM cells.  PARITY UNPINNED against MXNet / gluonnlp (absent); the backward pass is
torch autograd's.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch


def _gru_cell(x, h, wi, wh, bi, bh):
    H = h.shape[-1]
    gi = x @ wi.T + bi
    gh = h @ wh.T + bh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def _lstm_cell(x, h, c, wi, wh, bi, bh):
    H = h.shape[-1]
    g = x @ wi.T + bi + h @ wh.T + bh
    i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def _cell(cell, x, h, c, wi, wh, bi, bh):
    if cell == "lstm":
        return _lstm_cell(x, h, c, wi, wh, bi, bh)
    return _gru_cell(x, h, wi, wh, bi, bh), c


def _direction(x, w, pref, reverse, vl, cell="gru"):
    """gluon unroll(valid_length=...): steps past the valid length neither update the state nor emit output; the
    reverse direction starts at each row's last valid step (oracle/rnn_np.py::rnn_direction)."""
    B, T, _ = x.shape
    wi, wh, bi, bh = (w[pref + k] for k in ("i2h_weight", "h2h_weight", "i2h_bias", "h2h_bias"))
    H = wh.shape[1]
    h = torch.zeros((B, H), dtype=x.dtype)
    c = torch.zeros((B, H), dtype=x.dtype)
    outs = [torch.zeros((B, H), dtype=x.dtype) for _ in range(T)]
    ar = torch.arange(B)
    for s in range(T):
        idx = (vl - 1 - s) if reverse else torch.full((B,), s, dtype=torch.long)
        act = (s < vl) if reverse else (idx < vl)
        idc = idx.clamp(0, T - 1)
        hn, cn = _cell(cell, x[ar, idc], h, c, wi, wh, bi, bh)
        m = act[:, None].to(x.dtype)
        h = m * hn + (1 - m) * h
        c = m * cn + (1 - m) * c
        for b in range(B):
            if bool(act[b]):
                outs[int(idc[b])] = outs[int(idc[b])].clone()
                outs[int(idc[b])][b] = hn[b]
    return torch.stack(outs, dim=1), h, c


def forward_loss(params: dict, src, src_vl, tgt, tgt_vl, hidden, prefix="gnmt_", dtype=torch.float64, masks=None, cell="gru",
                 num_layers=2, num_bi_layers=1, use_residual=False):
    """-> (loss scalar tensor, logits (B, L-1, V), leaf tensors dict).

    ``num_layers`` / ``num_bi_layers`` / ``use_residual``: GNMTEncoder.forward (gnmt.py:136-160: dropout on every layer's output,
    ``outputs + inputs`` for layers ``i > num_bi_layers``, the backward direction's states of a bidirectional layer) and
    GNMTDecoder.hybrid_forward (gnmt.py:369-404: every cell behind the first reads ``[output of the layer below, attention]``, its
    output goes through dropout and, with ``use_residual``, gets the layer's input added).

    masks: the dropout masks (already scaled by 1/(1-p)); None = no dropout.  Two-layer form (rounds 2-3): a tuple
    ``(m_enc0 (B,T,2H), m_enc1 (B,T,H), m_dec (L,B,H))``; general form: a dict ``{"enc": [per encoder layer (B,T,dirs*H)],
    "dec": {j: (L,B,H) for decoder layers j >= 1}}``."""
    w = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
    x = torch.tensor(np.asarray(src), dtype=dtype)
    vl = torch.tensor(np.asarray(src_vl), dtype=torch.long)
    B, T, _ = x.shape
    H = hidden
    NL, NBI = num_layers, num_bi_layers
    if masks is not None and not isinstance(masks, dict):
        assert NL == 2
        masks = {"enc": [masks[0], masks[1]], "dec": {1: masks[2]}}
    tt = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    pe = prefix + "enc_"
    inputs = x
    h_init, c_init = [], []
    for i in range(NL):
        if i < NBI:
            fo, _, _ = _direction(inputs, w, f"{pe}rnn{i}_l_", False, vl, cell)
            bo, bh, bc = _direction(inputs, w, f"{pe}rnn{i}_r_", True, vl, cell)
            out = torch.cat([fo, bo], dim=2)
            h_init.append(bh); c_init.append(bc)
        else:
            out, hh, cc = _direction(inputs, w, f"{pe}rnn{i}_", False, vl, cell)
            h_init.append(hh); c_init.append(cc)
        if masks is not None:
            out = out * tt(masks["enc"][i])            # dropout on the layer output (states are not dropped)
        if use_residual and i > NBI:
            out = out + inputs
        inputs = out
    mem = inputs
    keyproj = mem @ w[prefix + "dec_attention_key_weight"].T
    mask = (torch.arange(T)[None, :] < vl[:, None])
    tg = torch.tensor(np.asarray(tgt), dtype=torch.long)
    tvl = torch.tensor(np.asarray(tgt_vl), dtype=torch.long) - 1
    L = tg.shape[1] - 1
    hs, cs = list(h_init), list(c_init)
    att = torch.zeros((B, H), dtype=dtype)
    pd = prefix + "dec_"
    cw = lambda j: (w[f"{pd}rnn{j}_i2h_weight"], w[f"{pd}rnn{j}_h2h_weight"], w[f"{pd}rnn{j}_i2h_bias"], w[f"{pd}rnn{j}_h2h_bias"])
    logits = []
    for i in range(L):
        emb = w[prefix + "tgt_embed_weight"][tg[:, i].clamp(min=0)]
        hs[0], cs[0] = _cell(cell, torch.cat([emb, att], dim=1), hs[0], cs[0], *cw(0))
        q = hs[0] / np.sqrt(H)
        score = torch.einsum("bh,bth->bt", q, keyproj)
        score = torch.where(mask, score, torch.full_like(score, -1e18))
        wts = torch.softmax(score, dim=1) * mask.to(dtype)
        att = torch.einsum("bt,bth->bh", wts, mem)
        rnn_out = hs[0]
        for j in range(1, NL):
            cur = rnn_out
            hs[j], cs[j] = _cell(cell, torch.cat([cur, att], dim=1), hs[j], cs[j], *cw(j))
            rnn_out = hs[j]
            if masks is not None:
                rnn_out = rnn_out * tt(masks["dec"][j][i])
            if use_residual:
                rnn_out = rnn_out + cur
        logits.append(rnn_out @ w[prefix + "tgt_proj_weight"].T + w[prefix + "tgt_proj_bias"])
    logits = torch.stack(logits, dim=1)                                       # (B, L, V)
    logp = torch.log_softmax(logits, dim=2)
    nll = -torch.gather(logp, 2, tg[:, 1:, None]).squeeze(2)
    m = (torch.arange(L)[None, :] < tvl[:, None]).to(dtype)
    per_sample = (nll * m).mean(dim=1)                                        # MaskedSoftmaxCELoss -> (B,)
    loss = per_sample.mean() * L / tvl.to(dtype).mean()
    return loss, logits, w


def loss_and_grads(params, src, src_vl, tgt, tgt_vl, hidden, prefix="gnmt_", masks=None, cell="gru", num_layers=2, num_bi_layers=1,
                   use_residual=False):
    loss, logits, w = forward_loss(params, src, src_vl, tgt, tgt_vl, hidden, prefix, masks=masks, cell=cell, num_layers=num_layers,
                                   num_bi_layers=num_bi_layers, use_residual=use_residual)
    loss.backward()
    return float(loss.detach()), logits.detach().numpy(), {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in w.items()}


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """MXNet Adam.update [EXT] for step count t (1-based); returns (params, m, v)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    np_, nm, nv = {}, {}, {}
    for k in p:
        nm[k] = beta1 * m.get(k, 0.0) + (1 - beta1) * g[k]
        nv[k] = beta2 * v.get(k, 0.0) + (1 - beta2) * g[k] * g[k]
        np_[k] = p[k] - lr_t * nm[k] / (np.sqrt(nv[k]) + eps)
    return np_, nm, nv
