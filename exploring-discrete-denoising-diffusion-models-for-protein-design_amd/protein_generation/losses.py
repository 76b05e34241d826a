"""LVB loss of the protein-style D3PM with the constructor and ``forward``
signatures of ``protein_generation/losses.py`` (:24-155), computed per batch in
one kernel launch (no Python loop over sequences, no [L,K,K] temporaries, no
host synchronisation on the timesteps).

``forward`` takes the reference's eight positional tensors.  The one-hot
arguments are redundant for the kernels: pass integer tokens through the
optional ``src=`` keyword (the corrupted sequence the collater also returns) to
skip the argmax that otherwise recovers them from ``src_onehot``."""
from __future__ import annotations

import numpy as np
import torch
from torch.nn import KLDivLoss

from .. import _lib
from .._ops import ops
from ..tables import protein_tables
from .constants import MSA_AAS
from .utils import Tokenizer


def sample_prior(a, b, len=len(MSA_AAS)):
    """Uniform prior [a,b] over `len` values (losses.py:6-13)."""
    return torch.ones(a, b) / len


def sample_priorMSA(a, b, c, len=len(MSA_AAS)):
    """Uniform prior [a,b,c] (losses.py:15-22)."""
    return torch.ones(a, b, c) / len


def KL_test(q, tokenizer):
    """KL(q || uniform), batchmean (losses.py:159-162)."""
    prior = sample_priorMSA(q.shape[0], q.shape[1], q.shape[2], len=tokenizer.K).to(q.device)
    return KLDivLoss(reduction="batchmean", log_target=False)(prior.log(), q)


class _LVB(torch.autograd.Function):
    """per_seq[B] (float64) and, in the same launch, d(baked * sum per_seq)/d pred."""

    @staticmethod
    def forward(ctx, pred, x0, x_t, t, q, tables, tmax, msa, baked):
        B = x0.shape[0]
        npos = x0.numel() // max(B, 1)
        pr = pred.contiguous()
        if pr.dtype != torch.float32:
            pr = pr.float()
        ld = pr.shape[-1]
        want_grad = ctx.needs_input_grad[0]
        per, dp = ops.lvb_loss(*tables.op_args(), pr, x0, x_t, t, q, int(tmax), int(msa), float(baked), want_grad)
        if not want_grad:
            dp = None
        ctx.dp, ctx.baked, ctx.shape, ctx.dtype = dp, float(np.float32(baked)), pred.shape, pred.dtype
        if want_grad:       # kept for a second backward (retain_graph): the kernel is run again
            ctx.again = (pr, x0, x_t, t, q, tables, int(tmax), int(msa))
        return per

    @staticmethod
    def backward(ctx, g):
        dp = ctx.dp
        if dp is None:
            return (None,) * 9
        g = g.contiguous().float()
        B = g.shape[0]
        baked = ctx.baked
        if getattr(ctx, "spent", False):
            # the first backward gave its buffer away (rescaled in place): recompute with unit scale
            pr, x0, x_t, t, q, tables, tmax, msa = ctx.again
            baked = 1.0
            with torch.cuda.device(pr.device):
                dp = ops.lvb_loss(*tables.op_args(), pr, x0, x_t, t, q, tmax, msa, 1.0, True)[1]
        ctx.spent = True
        with torch.cuda.device(dp.device):
            ops.scale_rows_(dp, g, baked)
        out = dp.view(ctx.shape)
        if out.dtype != ctx.dtype:
            out = out.to(ctx.dtype)
        return out, None, None, None, None, None, None, None, None


def _lvb_forward(self, src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, msa, src=None):
    _lib.require_device(predictions)
    dev = predictions.device
    K = self.K
    x0 = tgt.to(device=dev, dtype=torch.int64).contiguous()
    if src is None:
        src = src_onehot.to(dev)[..., :K].argmax(dim=-1)
    x_t = src.to(device=dev, dtype=torch.int64).contiguous()
    t = timestep.to(device=dev, dtype=torch.int64).contiguous()
    assert x_t.shape == x0.shape and t.shape == (x0.shape[0],)
    assert predictions.shape[:-1] == x0.shape and predictions.shape[-1] >= K
    qd = None
    if q is not None:
        # the reference only reads q[i] for i < B (losses.py:56-60, :128-132), so a leading size beyond B is
        # accepted: CollaterMSA returns q_x appended twice, [2B, D, L, K] (dataset.py:287,291)
        assert q.shape[1:] == x0.shape[1:] + (K,) and q.shape[0] >= x0.shape[0], (q.shape, x0.shape)
        qd = q[:x0.shape[0]].to(device=dev, dtype=torch.float64).contiguous()
    tables = protein_tables(Q, Q_bar, device=dev)
    B = x0.shape[0]
    baked = 1.0 if self.evaluation else float(np.float32(1.0) / np.float32(max(B, 1)))
    depth = int(x0.shape[1]) if msa else 0            # [B, D, L]: MSA depth
    with torch.cuda.device(dev):
        per = _LVB.apply(predictions, x0, x_t, t, qd, tables, self.tmax, depth, baked)
    if self.evaluation:
        return per                                   # losses.py:89-90
    return per.sum() / B                             # losses.py:92


class D3PMLVBLoss(KLDivLoss):
    """losses.py:24-93.  Inputs (names as in the reference's code, which swaps
    the roles its docstring gives): src_onehot [B,L,K] one-hot of the corrupted
    x_t, q [B,L,K] forward probabilities, predicitons [B,L,n_tokens] logits,
    tgt [B,L] clean tokens x_0, tgt_onehot, timestep [B], Q / Q_bar [T,K,K]."""

    def __init__(self, tmax=500, reduction="batchmean", log_target=False, tokenizer=None, evaluation=None):
        super().__init__(reduction=reduction, log_target=log_target)
        self.tmax = tmax
        self.tokenizer = tokenizer if tokenizer is not None else Tokenizer()
        self.K = self.tokenizer.K
        self.evaluation = evaluation

    def forward(self, src_onehot, q, predicitons, tgt, tgt_onehot, timestep, Q, Q_bar, src=None):
        return _lvb_forward(self, src_onehot, q, predicitons, tgt, tgt_onehot, timestep, Q, Q_bar, msa=False, src=src)


class D3PMLVBLossMSA(KLDivLoss):
    """losses.py:96-155: same over [B,D,L,...] inputs; the posterior is divided
    by Qbar_{t-1}[x0, x_t] as in the reference (:148)."""

    def __init__(self, tmax=500, reduction="batchmean", log_target=False, tokenizer=None):
        super().__init__(reduction=reduction, log_target=log_target)
        self.tmax = tmax
        self.tokenizer = tokenizer if tokenizer is not None else Tokenizer()
        self.K = self.tokenizer.K
        self.evaluation = None

    def forward(self, src_one_hot, q, predictions, tgt, tgt_one_hot, timestep, Q, Q_bar, src=None):
        return _lvb_forward(self, src_one_hot, q, predictions, tgt, tgt_one_hot, timestep, Q, Q_bar, msa=True, src=src)
