"""ORACLE (test infrastructure only) — fp32/fp64 CPU restatement of the temporal-head training step.

  model    reference models/vision/definitions.py:94-110 (CNNRNN, model=None: bi-GRU | bi-LSTM -> max over T -> Dense)
  loss     gluon.loss.SoftmaxCrossEntropyLoss, sparse labels, per sample            (train.py:324)  [EXT]
  backward ag.backward of the per-sample losses = gradient of their SUM             (train.py:419-421)
  update   gluon.Trainer 'sgd' .step(batch_size): rescale_grad = 1/batch_size;
           MXNet sgd_mom_update: mom = momentum*mom - lr*(rescale*grad + wd*w); w += mom   (train.py:298-299,424) [EXT]

PARITY UNPINNED against MXNet (absent); the backward pass is pinned to torch autograd on the CPU
(tests/test_cpu_oracle.py::test_train_oracle_matches_torch_autograd).  Gate order: GRU [r, z, n], LSTM [i, f, g, o]
(SURVEY App. B).
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _gru_dir(x, wi, wh, bi, bh, reverse):
    """-> seq (B,T,H) and the per-step cache."""
    B, T, _ = x.shape
    H = wh.shape[1]
    h = np.zeros((B, H), x.dtype)
    seq = np.zeros((B, T, H), x.dtype)
    cache = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = x[:, t] @ wi.T + bi
        gh = h @ wh.T + bh
        r = _sig(gi[:, :H] + gh[:, :H])
        z = _sig(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (1 - z) * n + z * h
        cache[t] = (h, r, z, n, gh[:, 2 * H:])
        seq[:, t] = hn
        h = hn
    return seq, cache


def _lstm_dir(x, wi, wh, bi, bh, reverse):
    B, T, _ = x.shape
    H = wh.shape[1]
    h = np.zeros((B, H), x.dtype)
    c = np.zeros((B, H), x.dtype)
    seq = np.zeros((B, T, H), x.dtype)
    cache = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        a = x[:, t] @ wi.T + bi + h @ wh.T + bh
        i, f, g, o = _sig(a[:, :H]), _sig(a[:, H:2 * H]), np.tanh(a[:, 2 * H:3 * H]), _sig(a[:, 3 * H:])
        c2 = f * c + i * g
        hn = o * np.tanh(c2)
        cache[t] = (h, c, i, f, g, o, c2)
        seq[:, t] = hn
        h, c = hn, c2
    return seq, cache


def forward_backward(x, labels, p, rnn_prefix=None, dense_prefix="cnnrnn0_dense0_", dtype=np.float64, cell="gru"):
    """-> (loss (B,), logits (B,C), grads dict with the parameter names)."""
    if rnn_prefix is None:
        rnn_prefix = f"cnnrnn0_{cell}0_"
    G = 3 if cell == "gru" else 4
    x = x.astype(dtype)
    q = {k: v.astype(dtype) for k, v in p.items()}
    B, T, F = x.shape
    seqs, caches = [], []
    for d, rev in (("l0_", False), ("r0_", True)):
        s, c = (_gru_dir if cell == "gru" else _lstm_dir)(x, q[rnn_prefix + d + "i2h_weight"], q[rnn_prefix + d + "h2h_weight"],
                        q[rnn_prefix + d + "i2h_bias"], q[rnn_prefix + d + "h2h_bias"], rev)
        seqs.append(s); caches.append(c)
    seq = np.concatenate(seqs, axis=2)                      # (B,T,2H)
    arg = seq.argmax(axis=1)                                # first maximum
    pooled = np.take_along_axis(seq, arg[:, None, :], axis=1)[:, 0]
    wd, bd = q[dense_prefix + "weight"], q[dense_prefix + "bias"]
    logits = pooled @ wd.T + bd
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    loss = lse - logits[np.arange(B), labels]
    dlog = np.exp(logits - lse[:, None])
    dlog[np.arange(B), labels] -= 1.0
    g = {dense_prefix + "weight": dlog.T @ pooled, dense_prefix + "bias": dlog.sum(axis=0)}
    dpooled = dlog @ wd
    dseq = np.zeros_like(seq)
    np.put_along_axis(dseq, arg[:, None, :], dpooled[:, None, :], axis=1)
    H = seq.shape[2] // 2
    for di, (d, rev) in enumerate((("l0_", False), ("r0_", True))):
        wi, wh = q[rnn_prefix + d + "i2h_weight"], q[rnn_prefix + d + "h2h_weight"]
        dwi, dwh = np.zeros_like(wi), np.zeros_like(wh)
        dbi, dbh = np.zeros(G * H, dtype), np.zeros(G * H, dtype)
        dh = np.zeros((B, H), dtype)
        dc = np.zeros((B, H), dtype)
        order = range(T) if rev else range(T - 1, -1, -1)   # reverse of the direction's walking order
        for t in order:
            dht = dh + dseq[:, t, di * H:(di + 1) * H]
            if cell == "lstm":
                hp, cp, i, f, gg, o, c2 = caches[di][t]
                tc = np.tanh(c2)
                dct = dc + dht * o * (1 - tc * tc)
                dgi = np.concatenate([dct * gg * i * (1 - i), dct * cp * f * (1 - f), dct * i * (1 - gg * gg),
                                      dht * tc * o * (1 - o)], axis=1)
                dwi += dgi.T @ x[:, t]; dbi += dgi.sum(axis=0)
                dwh += dgi.T @ hp; dbh += dgi.sum(axis=0)
                dh = dgi @ wh
                dc = dct * f
                continue
            hp, r, z, n, ghn = caches[di][t]
            dn = dht * (1 - z); dz = dht * (hp - n)
            dnp = dn * (1 - n * n); dzp = dz * z * (1 - z); drp = dnp * ghn * r * (1 - r)
            dgi = np.concatenate([drp, dzp, dnp], axis=1)
            dgh = np.concatenate([drp, dzp, dnp * r], axis=1)
            dwi += dgi.T @ x[:, t]; dbi += dgi.sum(axis=0)
            dwh += dgh.T @ hp; dbh += dgh.sum(axis=0)
            dh = dht * z + dgh @ wh
        g[rnn_prefix + d + "i2h_weight"], g[rnn_prefix + d + "h2h_weight"] = dwi, dwh
        g[rnn_prefix + d + "i2h_bias"], g[rnn_prefix + d + "h2h_bias"] = dbi, dbh
    return loss, logits, g


def sgd_momentum(p, g, mom, lr, momentum, wd, rescale):
    """In place on copies: returns (new params, new momentum)."""
    np_, nm = {}, {}
    for k in p:
        m = momentum * mom.get(k, 0.0) - lr * (rescale * g[k] + wd * p[k])
        nm[k] = m
        np_[k] = p[k] + m
    return np_, nm
