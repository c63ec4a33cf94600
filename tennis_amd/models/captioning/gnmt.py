"""Encoder and decoder of the captioner — mirror of reference models/captioning/gnmt.py
(itself derived from gluon-nlp) and of the NMTModel the reference builds from gluonnlp
(train_gnmt.py:228-229).  The blocks hold configuration and Gluon-named parameters; the
compute runs in libtennis_hip.so (tn_gnmt_*): cell_type 'gru' (flag default) or 'lstm', attention 'scaled_luong',
num_layers >= 2 with num_bi_layers < num_layers, use_residual on or off - inference (tn_gnmt_create_ex) and, since round 4, the
training step (tn_gnmt_trainer_create_ex).
"""
from __future__ import annotations

import numpy as np

from ... import weights as W
from ...block import Block, Parameter

__all__ = ["GNMTEncoder", "GNMTDecoder", "get_gnmt_encoder_decoder", "NMTModel", "Vocab", "TokenEmbedding"]


_CELL = ("i2h_weight", "h2h_weight", "i2h_bias", "h2h_bias")


def _cell_params(block, pref):
    for n in _CELL:
        block._own_params[pref + n] = Parameter(pref + n)


class GNMTEncoder(Block):
    """reference gnmt.py:30-160: num_bi_layers bidirectional layers, then uni-directional ones."""

    def __init__(self, cell_type="lstm", num_layers=2, num_bi_layers=1, hidden_size=128, dropout=0.0,
                 use_residual=True, prefix=None, **kwargs):
        super().__init__(prefix=prefix)
        assert num_bi_layers <= num_layers                                  # gnmt.py:78-80
        self._cell_type, self._num_layers, self._num_bi_layers = cell_type, num_layers, num_bi_layers
        self._hidden_size, self._dropout, self._use_residual = hidden_size, dropout, use_residual
        for i in range(num_layers):
            if i < num_bi_layers:
                _cell_params(self, f"{self.prefix}rnn{i}_l_")
                _cell_params(self, f"{self.prefix}rnn{i}_r_")
            else:
                _cell_params(self, f"{self.prefix}rnn{i}_")

    def _structural_params(self, path=""):
        """The block tree reference gnmt.py:84-111 builds: ``rnn_cells`` is a HybridSequential (children named by
        index) of ``BidirectionalCell(l_cell, r_cell)`` for i < num_bi_layers and plain cells after; a cell's own
        parameters are ``i2h_weight`` ... [EXT: Block._collect_params_with_prefix, BidirectionalCell.register_child]."""
        out = {}
        for i in range(self._num_layers):
            for n in _CELL:
                if i < self._num_bi_layers:
                    out[f"{path}rnn_cells.{i}.l_cell.{n}"] = f"{self.prefix}rnn{i}_l_{n}"
                    out[f"{path}rnn_cells.{i}.r_cell.{n}"] = f"{self.prefix}rnn{i}_r_{n}"
                else:
                    out[f"{path}rnn_cells.{i}.{n}"] = f"{self.prefix}rnn{i}_{n}"
        return out


class GNMTDecoder(Block):
    """reference gnmt.py:163-404."""

    def __init__(self, cell_type="lstm", attention_cell="scaled_luong", num_layers=2, hidden_size=128, dropout=0.0,
                 use_residual=True, output_attention=False, prefix=None, **kwargs):
        super().__init__(prefix=prefix)
        self._cell_type, self._num_layers, self._hidden_size = cell_type, num_layers, hidden_size
        self._attention_cell, self._use_residual, self._output_attention = attention_cell, use_residual, output_attention
        for i in range(num_layers):
            _cell_params(self, f"{self.prefix}rnn{i}_")
        self._own_params[self.prefix + "attention_key_weight"] = Parameter(self.prefix + "attention_key_weight")

    # reference gnmt.py:212-221: ``attention_cell`` (gluonnlp DotProductAttentionCell(units=H, scaled=True,
    # luong_style=True, use_bias=False)), ``dropout_layer``, ``rnn_cells`` HybridSequential of ``num_layers`` cells.
    # [EXT, gluonnlp attention_cell.py] In luong style the cell owns ONE bias-free Dense(H), held in the attribute
    # ``_proj_query`` and applied to the query: score = <W q, k> / sqrt(H).  The engine keeps the same bilinear form
    # with the matrix on the memory side (keys are projected once per clip, not once per step):
    # <W q, k> = <q, W^T k>, so attention_key_weight = W^T and the checkpoint array is stored transposed.
    # A file that names the matrix ``_proj_key.weight`` (score = <q, W k>) is accepted as is.
    _ATT_QUERY = "attention_cell._proj_query.weight"
    _ATT_KEY = "attention_cell._proj_key.weight"

    def _structural_params(self, path=""):
        out = {path + self._ATT_QUERY: self.prefix + "attention_key_weight",
               path + self._ATT_KEY: self.prefix + "attention_key_weight"}
        for i in range(self._num_layers):
            for n in _CELL:
                out[f"{path}rnn_cells.{i}.{n}"] = f"{self.prefix}rnn{i}_{n}"
        return out

    def _structural_transposed(self, path=""):
        return {path + self._ATT_QUERY}


def get_gnmt_encoder_decoder(cell_type="lstm", attention_cell="scaled_luong", num_layers=2, num_bi_layers=1,
                             hidden_size=128, dropout=0.0, use_residual=False, prefix="gnmt_", **kwargs):
    """reference gnmt.py:407-455 (same defaults)."""
    encoder = GNMTEncoder(cell_type=cell_type, num_layers=num_layers, num_bi_layers=num_bi_layers,
                          hidden_size=hidden_size, dropout=dropout, use_residual=use_residual, prefix=prefix + "enc_")
    decoder = GNMTDecoder(cell_type=cell_type, attention_cell=attention_cell, num_layers=num_layers,
                          hidden_size=hidden_size, dropout=dropout, use_residual=use_residual, prefix=prefix + "dec_")
    return encoder, decoder


class TokenEmbedding:
    """``gluonnlp.embedding.TokenEmbedding.from_file(file_path)`` (reference train_gnmt.py:212; the file is written by
    train_embeddings.py: ``data/embeddings-ex.txt``, 250 lines of ``token v1 ... v100``) [EXT]: one token per line followed by its
    vector, elements separated by ``elem_delim``; a first line of two elements is a header (word2vec / fastText) and skipped; a
    token seen twice keeps its first vector; a line whose vector length differs from the first one's is an error."""

    def __init__(self, idx_to_token, idx_to_vec):
        self.idx_to_token = list(idx_to_token)
        self.token_to_idx = {t: i for i, t in enumerate(self.idx_to_token)}
        self.idx_to_vec = np.ascontiguousarray(idx_to_vec, dtype=np.float32)
        self.dim = int(self.idx_to_vec.shape[1]) if self.idx_to_vec.ndim == 2 else 0

    def __len__(self):
        return len(self.idx_to_token)

    def __contains__(self, token):
        return token in self.token_to_idx

    def __getitem__(self, token):
        if isinstance(token, str):
            j = self.token_to_idx.get(token)
            return self.idx_to_vec[j] if j is not None else np.zeros(self.dim, np.float32)
        return np.stack([self[t] for t in token])

    @classmethod
    def from_file(cls, file_path, elem_delim=" ", encoding="utf8"):
        toks, vecs, seen, dim = [], [], set(), None
        with open(file_path, "r", encoding=encoding) as f:
            for line_num, line in enumerate(f):
                elems = line.rstrip().split(elem_delim)
                if len(elems) < 2 or (line_num == 0 and len(elems) == 2):
                    continue                      # blank line / "<count> <dim>" header
                token, vals = elems[0], [e for e in elems[1:] if e != ""]
                if token in seen:
                    continue
                vec = np.array([float(v) for v in vals], np.float32)
                if dim is None:
                    dim = vec.size
                elif vec.size != dim:
                    raise ValueError(f"{file_path}:{line_num + 1}: vector of {vec.size} elements, the file's first one has {dim}")
                seen.add(token)
                toks.append(token)
                vecs.append(vec)
        if not toks:
            raise ValueError(f"{file_path}: no embedding vectors")
        return cls(toks, np.stack(vecs))


class Vocab:
    """gluonnlp.Vocab(counter) as the reference uses it (dataset.py:57-58): indices 0..3 are
    <unk>, <pad>, <bos>, <eos>; then tokens by descending frequency, ties alphabetical [EXT]."""

    def __init__(self, counter: dict):
        self.unknown_token, self.padding_token, self.bos_token, self.eos_token = "<unk>", "<pad>", "<bos>", "<eos>"
        toks = sorted(counter.items(), key=lambda kv: (-kv[1], kv[0]))
        self.idx_to_token = [self.unknown_token, self.padding_token, self.bos_token, self.eos_token] + [t for t, _ in toks]
        self.token_to_idx = {t: i for i, t in enumerate(self.idx_to_token)}
        self.embedding = None

    def __len__(self):
        return len(self.idx_to_token)

    def set_embedding(self, *embeddings):
        """``gluonnlp.Vocab.set_embedding(word_embs)`` as the reference calls it (train_gnmt.py:211-213) [EXT]: ``self.embedding``
        gets ``idx_to_vec`` (len(vocab), dim): the file's vector for every vocabulary token the file holds, the embedding's
        unknown vector - zeros unless the file itself defines ``<unk>`` - for every other one, the special tokens included
        (SURVEY App. B); several embeddings are concatenated along the vector."""
        if not embeddings or any(e is None for e in embeddings):
            self.embedding = None
            return
        cols = []
        for e in embeddings:
            unk = e.idx_to_vec[e.token_to_idx[self.unknown_token]] if self.unknown_token in e.token_to_idx else np.zeros(e.dim, np.float32)
            tab = np.tile(unk.astype(np.float32), (len(self), 1))
            for i, t in enumerate(self.idx_to_token):
                j = e.token_to_idx.get(t)
                if j is not None:
                    tab[i] = e.idx_to_vec[j]
            cols.append(tab)
        self.embedding = TokenEmbedding(list(self.idx_to_token), np.concatenate(cols, axis=1))

    def __getitem__(self, tokens):
        if isinstance(tokens, str):
            return self.token_to_idx.get(tokens, 0)
        return [self.token_to_idx.get(t, 0) for t in tokens]


class NMTModel(Block):
    """``gluonnlp.model.translation.NMTModel(src_vocab=None, tgt_vocab, encoder, decoder, embed_size,
    prefix, src_embed, tgt_embed)`` as called at reference train_gnmt.py:228-229: Embedding(V, embed) target,
    Dense(V) projection, and a source "embedding" that is either the identity (feature mode: pre-extracted frame
    features, train_gnmt.py:188-192) or ``TimeDistributed(backbone)`` (frame mode, train_gnmt.py:148-170: the clip's
    frames go through the CNN inside the model).  ``src_embed``: None / identity, or a block mapping
    (B, T, frame...) -> (B, T, input_size)."""

    def __init__(self, src_vocab=None, tgt_vocab=None, encoder=None, decoder=None, embed_size=100, prefix="gnmt_",
                 src_embed=None, tgt_embed=None, input_size=1024, seed=7, **kwargs):
        super().__init__(prefix=prefix)
        self.tgt_vocab, self.encoder, self.decoder = tgt_vocab, encoder, decoder
        self._embed_size, self._input_size, self._seed = embed_size, input_size, seed
        self.src_embed = src_embed if isinstance(src_embed, Block) else None
        for n in ("tgt_proj_weight", "tgt_proj_bias", "tgt_embed_weight"):
            self._own_params[prefix + n] = Parameter(prefix + n)
        self._tgt_embed_given = tgt_embed is not None
        if tgt_embed is not None:                      # train_gnmt.py:211-218: preloaded embedding table (its width IS the embed size)
            tab = np.ascontiguousarray(getattr(tgt_embed, "idx_to_vec", tgt_embed), dtype=np.float32)
            if tab.ndim != 2 or (tgt_vocab is not None and tab.shape[0] != len(tgt_vocab)):
                raise ValueError(f"tgt_embed: a (len(tgt_vocab), dim) table is expected, got {tab.shape}")
            self._embed_size = int(tab.shape[1])
            self._own_params[prefix + "tgt_embed_weight"].data = tab

    def _structural_params(self, path=""):
        """[EXT, gluonnlp NMTModel.__init__] children ``src_embed``, ``tgt_embed``, ``encoder``, ``decoder``,
        ``tgt_proj`` (Dense).  ``tgt_embed`` is a HybridSequential(Embedding, Dropout) when the model builds it
        (``tgt_embed.0.weight``) and the caller's ``nn.Embedding`` itself when one is passed (train_gnmt.py:211-218:
        ``tgt_embed.weight``); both names are read, the one this model was built with is written."""
        emb = ["tgt_embed.weight", "tgt_embed.0.weight"] if self._tgt_embed_given else ["tgt_embed.0.weight", "tgt_embed.weight"]
        out = {path + n: self.prefix + "tgt_embed_weight" for n in emb}
        out[path + "tgt_proj.weight"] = self.prefix + "tgt_proj_weight"
        out[path + "tgt_proj.bias"] = self.prefix + "tgt_proj_bias"
        for name, child in self._children.items():      # encoder, decoder, src_embed (TimeDistributed(backbone))
            out.update(child._structural_params(path + name + "."))
        return out

    def initialize(self, init=None, ctx=None, **kwargs):
        super().initialize()
        if any(v.data is None for v in self.collect_params().values()):
            enc = self.encoder
            p = W.make_gnmt_weights(self._seed, enc._cell_type, self._input_size, enc._hidden_size, self._embed_size,
                                    len(self.tgt_vocab), enc._num_layers, enc._num_bi_layers, self.prefix)
            have = {k: v.data for k, v in self.collect_params().items() if v.data is not None}
            p.update(have)
            self.set_params(p)

    def embed_source(self, src):
        """``self.src_embed(src)`` of NMTModel.encode [EXT]: frames (B, T, 3, H, W) fp32 / (B, T, H, W, 3) fp16|u8 ->
        per-frame features (B, T, input_size) through the HIP frame encoder; features pass through unchanged."""
        if self.src_embed is None or getattr(src, "ndim", 0) != 5:
            return src
        feats = self.src_embed(src)
        if feats.shape[-1] != self._input_size:
            raise ValueError(f"src_embed yields {feats.shape[-1]}-d frame features, the encoder was built for {self._input_size}")
        return feats

    def _captioner(self, beam, max_length, max_batch, max_src_len):
        from ...engine import GNMTCaptioner
        key = (beam, max_length)
        if self._engine is None or self._engine[0] != key or self._engine[2] < max_batch or self._engine[3] < max_src_len:
            p = {k: v.data for k, v in self.collect_params().items()}
            enc = self.encoder
            if bool(enc._use_residual) != bool(self.decoder._use_residual) or enc._num_layers != self.decoder._num_layers:
                raise ValueError("encoder and decoder must agree on num_layers and use_residual (get_gnmt_encoder_decoder, gnmt.py:397-416)")
            cap = GNMTCaptioner(p, self._input_size, enc._hidden_size, self._embed_size, len(self.tgt_vocab), beam,
                                max_length, max(max_batch, 32), max(max_src_len, 256), self.prefix, enc._cell_type,
                                enc._num_layers, enc._num_bi_layers, bool(enc._use_residual))
            self._engine = (key, cap, max(max_batch, 32), max(max_src_len, 256))
        return self._engine[1]
