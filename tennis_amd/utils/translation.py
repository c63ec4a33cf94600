"""Beam-search translator — mirror of reference utils/translation.py::BeamSearchTranslator
and of gluonnlp's BeamSearchScorer as constructed at reference train_gnmt.py:250-252."""
from __future__ import annotations

import numpy as np
import torch

__all__ = ["BeamSearchTranslator", "BeamSearchScorer"]


class BeamSearchScorer:
    """``gluonnlp.model.BeamSearchScorer(alpha, K)``: LP(n) = ((K+n)/(K+1))**alpha [EXT]."""

    def __init__(self, alpha=1.0, K=5.0):
        self._alpha, self._K = float(alpha), float(K)


class BeamSearchTranslator:
    """reference utils/translation.py:28-82."""

    def __init__(self, model, beam_size=1, scorer=None, max_length=100):
        self._model = model
        self._beam_size = beam_size
        self._scorer = scorer if scorer is not None else BeamSearchScorer()
        self._max_length = max_length
        self._eos_id = model.tgt_vocab.token_to_idx[model.tgt_vocab.eos_token]        # translation.py:47
        self._bos_id = model.tgt_vocab.token_to_idx[model.tgt_vocab.bos_token]        # translation.py:79-80

    def translate(self, src_seq, src_valid_length):
        """translation.py:55-82 -> (samples (B,beam,L) int32, scores (B,beam) descending,
        valid_length (B,beam) int32) as torch CUDA tensors."""
        src = torch.as_tensor(np.asarray(src_seq)) if not isinstance(src_seq, torch.Tensor) else src_seq
        vl = torch.as_tensor(np.asarray(src_valid_length)) if not isinstance(src_valid_length, torch.Tensor) \
            else src_valid_length
        src = src.cuda() if not src.is_cuda else src
        src = self._model.embed_source(src)          # frame mode: TimeDistributed(backbone) (train_gnmt.py:168-170)
        b, t = src.shape[0], src.shape[1]
        cap = self._model._captioner(self._beam_size, self._max_length, b, t)
        cap.encode(src, vl.to(src.device))                                            # translation.py:76-78
        return cap.beam_search(self._bos_id, self._eos_id, self._scorer._alpha, self._scorer._K, self._max_length)
