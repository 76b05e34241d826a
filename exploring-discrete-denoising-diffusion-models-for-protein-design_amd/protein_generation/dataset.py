"""Batch collaters of the protein-style process with the names, arguments and
return tuples of ``protein_generation/dataset.py`` (:196-300).

The reference corrupts one sequence at a time on the host (a one-hot [L,K]
float64 matrix product and a ``torch.multinomial`` per sequence).  Here the
strings are tokenised on the host, moved to the device once as integers, and
the whole batch is corrupted by one kernel launch against the resident tables;
the one-hot and probability tensors of the reference's return tuple are
produced on the device, or skipped altogether with ``lean=True`` (the fused
loss needs only tokens and timesteps).

The dataset classes of the reference file (A2M parsing of one training MSA)
are input plumbing outside this package.
"""
from __future__ import annotations

import numpy as np
import torch

from ..tables import protein_tables
from .utils import Tokenizer, corrupt_batch, _default_device


def _tokens(tokenizer, sequences, device):
    """Equal-length strings -> int64 [B, L] on the device (one upload)."""
    tok = np.stack([tokenizer.tokenize(s) for s in sequences]).astype(np.int64)
    return torch.from_numpy(tok).to(device)


class Collater(object):
    """dataset.py:196-250: per sequence t ~ randint(1, num_steps) (numpy's global
    RNG, one draw per sequence in order, as the reference), x_t ~ q(x_t | x_0).

    __call__(data, noise=None) with data = [(sequence, name), ...] returns
    (src, src_one_hot, timesteps, tokenized, one_hot, Q, Q_bar, q_x, names), or
    (src, timesteps, tokenized) when ``lean``.  ``noise`` [B,L,K] float64 replays
    the Exp(1) draws of the reference's multinomial."""

    def __init__(self, tokenizer=None, num_steps=500, Q=None, Q_bar=None, device=None, lean=False):
        self.tokenizer = tokenizer or Tokenizer()
        self.num_steps = num_steps
        self.K = self.tokenizer.K
        self.Q = Q
        self.Q_bar = Q_bar
        self.device = torch.device(device) if device is not None else None
        self.lean = lean

    def _dev(self):
        if self.device is not None:
            return self.device
        return self.Q_bar.device if self.Q_bar.is_cuda else _default_device()

    def __call__(self, data, noise=None):
        sequences = [x[0] for x in data]
        names = [x[1] for x in data]
        assert len(sequences) == len(names)
        out = self.collate_tokens(_tokens(self.tokenizer, sequences, self._dev()), noise)
        return out if self.lean else out + (names,)

    def collate_tokens(self, tokenized, noise=None):
        """The same for sequences that are already tokenised ([B, L] int64, e.g. a data set kept on the device):
        the tuple of ``__call__`` without the names."""
        dev = self._dev()
        tokenized = tokenized.to(device=dev, dtype=torch.int64)
        timesteps = torch.tensor([np.random.randint(1, self.num_steps) for _ in range(tokenized.shape[0])],
                                 dtype=torch.long, device=dev)
        tables = protein_tables(self.Q, self.Q_bar, device=dev)
        src, q_x = corrupt_batch(tables, tokenized, timesteps, noise, want_probs=not self.lean)
        if self.lean:
            return src, timesteps, tokenized
        one_hot = self.tokenizer.one_hot(tokenized)
        src_one_hot = self.tokenizer.one_hot(src)
        return src, src_one_hot, timesteps, tokenized, one_hot, self.Q, self.Q_bar, q_x


class CollaterMSA(object):
    """dataset.py:253-300: one timestep per MSA, every row of the MSA corrupted
    with it.  __call__(msas, noise=None) with msas = [[row strings], ...] returns
    (src [B,D,L], src_one_hot, timesteps [B], tokenized [B,D,L], tgt_one_hot, Q,
    Q_bar, q_x).  The reference appends each MSA's probabilities twice
    (dataset.py:287,291), so its q_x is [2B,D,L,K] with every entry repeated;
    ``duplicate_q=True`` (the default) keeps that shape."""

    def __init__(self, tokenizer=None, num_steps=100, Q=None, Q_bar=None, num_seqs=64, device=None, lean=False,
                 duplicate_q=True):
        self.tokenizer = tokenizer or Tokenizer()
        self.num_steps = num_steps
        self.K = self.tokenizer.K
        self.Q = Q
        self.Q_bar = Q_bar
        self.num_seqs = num_seqs
        self.device = torch.device(device) if device is not None else None
        self.lean = lean
        self.duplicate_q = duplicate_q

    def __call__(self, msas, noise=None):
        dev = self.device if self.device is not None else (self.Q_bar.device if self.Q_bar.is_cuda else _default_device())
        tokenized = torch.stack([_tokens(self.tokenizer, msa, dev) for msa in msas])            # [B, D, L]
        B, D, L = tokenized.shape
        timesteps = torch.tensor([np.random.randint(1, self.num_steps) for _ in msas], dtype=torch.long, device=dev)
        tables = protein_tables(self.Q, self.Q_bar, device=dev)
        src, q_x = corrupt_batch(tables, tokenized.reshape(B, D * L), timesteps,
                                 None if noise is None else noise.reshape(B, D * L, self.K), want_probs=not self.lean)
        src = src.reshape(B, D, L)
        if self.lean:
            return src, timesteps, tokenized
        q_x = q_x.reshape(B, D, L, self.K)
        if self.duplicate_q:
            q_x = q_x.repeat_interleave(2, dim=0)
        return (src, self.tokenizer.one_hot(src), timesteps, tokenized, self.tokenizer.one_hot(tokenized), self.Q,
                self.Q_bar, q_x)
