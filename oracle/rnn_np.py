"""ORACLE (test infrastructure only) — fp32 CPU restatement of the recurrent ops.

PARITY UNPINNED (see oracle/densenet_np.py header): ``mx.gluon.rnn.GRU/LSTM``
and the Gluon cell classes are third-party, absent here; equations and gate
order follow the published cuDNN/MXNet convention [EXT, SURVEY App. B] and are
cross-checked against ``torch.nn.GRU/LSTM`` (identical convention) in
tests/test_cpu_oracle.py.

  GRU   gates [r, z, n]:  r = s(Wir x + bir + Whr h + bhr)
                          z = s(Wiz x + biz + Whz h + bhz)
                          n = tanh(Win x + bin + r * (Whn h + bhn))
                          h' = (1 - z) * n + z * h
  LSTM  gates [i, f, g, o]: c' = f*c + i*g ; h' = o * tanh(c')
"""
from __future__ import annotations

import numpy as np


def _sig(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(np.float32)


def gru_cell(x, h, wi, wh, bi, bh):
    hid = h.shape[-1]
    gi = x @ wi.T + bi
    gh = h @ wh.T + bh
    r = _sig(gi[:, :hid] + gh[:, :hid])
    z = _sig(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
    n = np.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:]).astype(np.float32)
    return ((1 - z) * n + z * h).astype(np.float32)


def lstm_cell(x, h, c, wi, wh, bi, bh):
    hid = h.shape[-1]
    g = x @ wi.T + bi + h @ wh.T + bh
    i = _sig(g[:, :hid])
    f = _sig(g[:, hid:2 * hid])
    gg = np.tanh(g[:, 2 * hid:3 * hid]).astype(np.float32)
    o = _sig(g[:, 3 * hid:])
    c2 = (f * c + i * gg).astype(np.float32)
    return (o * np.tanh(c2)).astype(np.float32), c2


def rnn_direction(x, p, pref, mode, reverse=False, valid_length=None, h0=None, c0=None):
    """One direction over (B,T,F).  With ``valid_length`` the reverse direction
    starts at each row's last valid step (``SequenceReverse`` with lengths) and
    steps past the valid length neither update the state nor emit output
    (Gluon ``unroll(valid_length=...)`` [EXT]; reference gnmt.py:141-143).
    Returns (out (B,T,H), h_last, c_last)."""
    b, t, _ = x.shape
    wi, wh = p[pref + "i2h_weight"], p[pref + "h2h_weight"]
    bi, bh = p[pref + "i2h_bias"], p[pref + "h2h_bias"]
    hid = wh.shape[1]
    h = np.zeros((b, hid), np.float32) if h0 is None else h0.copy()
    c = np.zeros((b, hid), np.float32) if c0 is None else c0.copy()
    out = np.zeros((b, t, hid), np.float32)
    vl = np.full(b, t, np.int64) if valid_length is None else np.asarray(valid_length).astype(np.int64)
    for s in range(t):
        # per-row time index: forward s; reverse vl-1-s (padding stays at the tail)
        idx = (vl - 1 - s) if reverse else np.full(b, s, np.int64)
        act = (s < vl) if reverse else (idx < vl)
        idx_c = np.clip(idx, 0, t - 1)
        xt = x[np.arange(b), idx_c]
        if mode == "gru":
            hn = gru_cell(xt, h, wi, wh, bi, bh)
            cn = c
        else:
            hn, cn = lstm_cell(xt, h, c, wi, wh, bi, bh)
        h = np.where(act[:, None], hn, h)
        c = np.where(act[:, None], cn, c)
        rows = np.nonzero(act)[0]
        out[rows, idx_c[rows]] = hn[rows]
    return out, h, c


def birnn_layer(x, p, prefix, mode, valid_length=None):
    """``mx.gluon.rnn.GRU/LSTM(H, layout='NTC', bidirectional=True)`` with zero
    initial state (reference definitions.py:94-96,106): out = concat(fwd,bwd)."""
    fo, fh, fc = rnn_direction(x, p, prefix + "l0_", mode, False, valid_length)
    bo, bh_, bc = rnn_direction(x, p, prefix + "r0_", mode, True, valid_length)
    return np.concatenate([fo, bo], axis=-1), (fh, fc), (bh_, bc)
