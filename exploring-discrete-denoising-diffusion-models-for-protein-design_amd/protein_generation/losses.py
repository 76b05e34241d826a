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
from ..d3pm.diffusion_discrete import _upstream


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
    """comb[B] = per_seq + lam * ce_seq (float64; lam None: per_seq alone) and, in the same launch,
    d(baked * sum comb)/d pred.  Second output: ce_seq (no gradient flows through it; None without lam)."""

    @staticmethod
    def _run(pr, x0, x_t, t, q, tables, tmax, msa, baked, lam, want_grad):
        if lam is None:
            per, dp = ops.lvb_loss(*tables.op_args(), pr, x0, x_t, t, q, tmax, msa, baked, want_grad)
            return per, None, dp
        per, ce, dp = ops.lvb_hybrid_loss(*tables.op_args(), pr, x0, x_t, t, q, tmax, msa, baked, baked * lam, want_grad)
        return (per + lam * ce if lam != 0.0 else per), ce, dp

    @staticmethod
    def forward(ctx, pred, x0, x_t, t, q, tables, tmax, msa, baked, lam):
        pr = pred.contiguous()
        if pr.dtype != torch.float32:
            pr = pr.float()
        want_grad = ctx.needs_input_grad[0]
        comb, ce, dp = _LVB._run(pr, x0, x_t, t, q, tables, int(tmax), int(msa), float(baked), lam, want_grad)
        if not want_grad:
            dp = None
        ctx.dp, ctx.baked, ctx.shape, ctx.dtype = dp, float(np.float32(baked)), pred.shape, pred.dtype
        if want_grad:       # kept for a second backward (retain_graph): the kernel is run again
            ctx.again = (pr, x0, x_t, t, q, tables, int(tmax), int(msa), lam)
        if ce is None:
            ce = comb.new_empty(0)
        ctx.mark_non_differentiable(ce)
        return comb, ce

    @staticmethod
    def backward(ctx, g, _g_ce):
        dp = ctx.dp
        if dp is None:
            return (None,) * 10
        g = _upstream(g)
        baked = ctx.baked
        if getattr(ctx, "spent", False):
            # the first backward gave its buffer away (rescaled in place): recompute with unit scale
            pr, x0, x_t, t, q, tables, tmax, msa, lam = ctx.again
            baked = 1.0
            dp = _LVB._run(pr, x0, x_t, t, q, tables, tmax, msa, 1.0, lam, True)[2]
        ctx.spent = True
        ops.scale_rows_(dp, g, baked)
        out = dp.view(ctx.shape)
        if out.dtype != ctx.dtype:
            out = out.to(ctx.dtype)
        return (out,) + (None,) * 9


def _lvb_forward(self, src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, msa, src=None, out_scale=None):
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
    scale = 1.0 if out_scale is None else float(out_scale)
    baked = scale if self.evaluation else float(np.float32(scale) / np.float32(max(B, 1)))
    depth = int(x0.shape[1]) if msa else 0            # [B, D, L]: MSA depth
    lam = None if self.hybrid_coeff is None else float(self.hybrid_coeff)
    comb, ce = _LVB.apply(predictions, x0, x_t, t, qd, tables, self.tmax, depth, baked, lam)
    self.ce_seq = ce if lam is not None else None
    if self.evaluation:
        return comb if out_scale is None else comb * scale                # losses.py:89-90
    if out_scale is None:
        return comb.sum() / B                                             # losses.py:92
    return comb.sum() * baked            # the upstream gradient of a plain .backward() equals the baked scale


class _HybridMixin:
    """``hybrid_coeff`` / ``out_scale``: the training loss as the reference's driver forms it around the module
    (Protein Generation Notebook cell-12: ``loss = (lvb + _lambda * ce) * n_tokens`` with ``ce`` the mean
    ``CrossEntropyLoss`` over every token of the batch and all ``n_tokens`` logits), computed by the same launch.
    ``hybrid_coeff=None`` (default) is the reference module: the LVB alone.  With a number (0 included) ``forward``
    returns ``lvb + hybrid_coeff * ce`` and ``ce_loss`` holds the batch's cross-entropy (for ``nll_loss = ce *
    n_tokens``); ``out_scale=`` in ``forward`` multiplies the result (and bakes the factor into the gradient, so
    that ``.backward()`` launches nothing further)."""

    hybrid_coeff = None
    ce_seq = None

    @property
    def ce_loss(self):
        """Mean cross-entropy over all tokens of the last batch (None unless ``hybrid_coeff`` is set)."""
        return None if self.ce_seq is None else self.ce_seq.mean()


class D3PMLVBLoss(_HybridMixin, KLDivLoss):
    """losses.py:24-93.  Inputs (names as in the reference's code, which swaps
    the roles its docstring gives): src_onehot [B,L,K] one-hot of the corrupted
    x_t, q [B,L,K] forward probabilities, predicitons [B,L,n_tokens] logits,
    tgt [B,L] clean tokens x_0, tgt_onehot, timestep [B], Q / Q_bar [T,K,K]."""

    def __init__(self, tmax=500, reduction="batchmean", log_target=False, tokenizer=None, evaluation=None,
                 hybrid_coeff=None):
        super().__init__(reduction=reduction, log_target=log_target)
        self.tmax = tmax
        self.tokenizer = tokenizer if tokenizer is not None else Tokenizer()
        self.K = self.tokenizer.K
        self.evaluation = evaluation
        self.hybrid_coeff = hybrid_coeff

    def forward(self, src_onehot, q, predicitons, tgt, tgt_onehot, timestep, Q, Q_bar, src=None, out_scale=None):
        return _lvb_forward(self, src_onehot, q, predicitons, tgt, tgt_onehot, timestep, Q, Q_bar, msa=False, src=src,
                            out_scale=out_scale)


class D3PMLVBLossMSA(_HybridMixin, KLDivLoss):
    """losses.py:96-155: same over [B,D,L,...] inputs; the posterior is divided
    by Qbar_{t-1}[x0, x_t] as in the reference (:148)."""

    def __init__(self, tmax=500, reduction="batchmean", log_target=False, tokenizer=None, hybrid_coeff=None):
        super().__init__(reduction=reduction, log_target=log_target)
        self.tmax = tmax
        self.tokenizer = tokenizer if tokenizer is not None else Tokenizer()
        self.K = self.tokenizer.K
        self.evaluation = None
        self.hybrid_coeff = hybrid_coeff

    def forward(self, src_one_hot, q, predictions, tgt, tgt_one_hot, timestep, Q, Q_bar, src=None, out_scale=None):
        return _lvb_forward(self, src_one_hot, q, predictions, tgt, tgt_one_hot, timestep, Q, Q_bar, msa=True, src=src,
                            out_scale=out_scale)
