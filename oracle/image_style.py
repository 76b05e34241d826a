"""Oracle (test infrastructure only) for the image-style D3PM process object.

Functional restatement of ``d3pm/diffusion_discrete.py`` (class
``DiscreteDiffusion``) and ``d3pm/utils.py`` in eager PyTorch-CPU ops.  The op
order and dtypes follow the reference so that results are bit-identical to it
on the same torch build: float64 tables, a float32 cast of the *whole* table
per gather, float32 everywhere after.

State is a plain ``Process`` record instead of a class with methods; every
function cites the reference lines it restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch
import torch.nn.functional as F

EPS = 1.0e-6  # diffusion_discrete.py:49


# --------------------------------------------------------------------------
# schedules and tables
# --------------------------------------------------------------------------
def make_betas(start: float, stop: float, num_steps: int, kind: str) -> torch.Tensor:
    """diffusion_discrete.py:12-26 (``generate_betas``)."""
    if kind == "linear":
        return torch.tensor(np.linspace(start, stop, num_steps))
    if kind == "cosine":
        s = np.arange(num_steps + 1, dtype=np.float64) / num_steps
        abar = np.cos((s + 0.008) / 1.008 * np.pi / 2)
        return torch.tensor(np.minimum(1 - abar[1:] / abar[:-1], 0.999))
    if kind == "jsd":
        return torch.tensor(1.0 / np.linspace(num_steps, 1.0, num_steps))
    raise NotImplementedError(kind)


def _onestep_uniform(beta_t: torch.Tensor, K: int, bands: Optional[int]) -> torch.Tensor:
    """diffusion_discrete.py:101-154."""
    if bands is None:
        m = torch.full(size=(K, K), fill_value=beta_t / float(K), dtype=torch.float64)
        m.fill_diagonal_(1.0 - beta_t * (K - 1.0) / K)
        return m
    m = torch.zeros((K, K), dtype=torch.float64)
    off = torch.full(size=(K - 1,), fill_value=beta_t / float(K), dtype=torch.float64)
    for k in range(1, bands + 1):
        m += torch.diag(off, diagonal=k)
        m += torch.diag(off, diagonal=-k)
        off = off[:-1]
    m += torch.diag(1.0 - m.sum(dim=1), diagonal=0)
    return m


def _onestep_gaussian(beta_t: torch.Tensor, K: int, bands: Optional[int]) -> torch.Tensor:
    """diffusion_discrete.py:156-190 (the 0..255 grid is literal, :166)."""
    nb = bands if bands else K - 1
    m = torch.zeros((K, K), dtype=torch.float64)
    v = torch.linspace(start=0.0, end=255.0, steps=K, dtype=torch.float64)
    v = v * 2.0 / (K - 1.0)
    v = v[: nb + 1]
    v = -v * v / beta_t
    v = torch.cat([v.flip(dims=[0]), v], dim=0)
    v = F.softmax(v, dim=0)
    v = v[nb:]
    for k in range(1, nb + 1):
        off = torch.full(size=(K - k,), fill_value=v[k], dtype=torch.float64)
        m += torch.diag(off, diagonal=k)
        m += torch.diag(off, diagonal=-k)
    m += torch.diag(1.0 - m.sum(dim=1), diagonal=0)
    return m


def _onestep_absorbing(beta_t: torch.Tensor, K: int) -> torch.Tensor:
    """diffusion_discrete.py:192-202.  The reference line :195 raises
    (``torch.full(shape=...)``); this is its evident intent: diag(1-beta) with
    beta added to column K//2."""
    m = torch.diag(torch.full(size=(K,), fill_value=1.0 - beta_t, dtype=torch.float64), diagonal=0)
    m[:, K // 2] += beta_t
    return m


@dataclass
class Process:
    """The attributes ``DiscreteDiffusion.__init__`` leaves behind (:36-99)."""

    betas: torch.Tensor            # [T] f64
    kind: str                      # uniform | gaussian | absorbing
    K: int
    bands: Optional[int]
    q_mats: torch.Tensor           # [T,K,K] f64, cumulative products
    qT_onestep: torch.Tensor       # [T,K,K] f64, transposed one-step (a view)
    hybrid_coeff: float = 0.0
    loss_type: str = "hybrid"
    eps: float = EPS

    @property
    def T(self) -> int:
        return self.betas.shape[0]


def build_process(betas: torch.Tensor, kind: str, num_bits: int, bands: Optional[int] = None,
                  hybrid_coeff: float = 0.0, loss_type: str = "hybrid") -> Process:
    """diffusion_discrete.py:51-99: validation, one-step list, numpy
    ``tensordot`` chain for the cumulative products, transposed view."""
    if not isinstance(betas, torch.Tensor):
        raise ValueError("expected betas to be a torch Tensor")
    if not ((betas > 0).all() and (betas <= 1).all()):
        raise ValueError("betas must be in (0, 1]")
    betas = betas.to(torch.float64)
    K = 2 ** num_bits
    T = betas.shape[0]
    if kind == "uniform":
        steps = [_onestep_uniform(betas[t], K, bands) for t in range(T)]
    elif kind == "gaussian":
        steps = [_onestep_gaussian(betas[t], K, bands) for t in range(T)]
    elif kind == "absorbing":
        steps = [_onestep_absorbing(betas[t], K) for t in range(T)]
    else:
        raise ValueError(f"transition_mat_type must be 'uniform', but is {kind}")
    onestep = torch.stack(steps, dim=0)
    cur = onestep[0].numpy()
    chain = [cur]
    for t in range(1, T):
        cur = np.tensordot(cur, onestep[t].numpy(), axes=[[1], [0]])
        chain.append(cur)
    q_mats = torch.tensor(np.stack(chain, axis=0))
    return Process(betas=betas, kind=kind, K=K, bands=bands, q_mats=q_mats,
                   qT_onestep=onestep.permute(0, 2, 1), hybrid_coeff=hybrid_coeff,
                   loss_type=loss_type)


# --------------------------------------------------------------------------
# gathers
# --------------------------------------------------------------------------
# Arithmetic type after the gathers: float32, as in the reference.  Tests switch it to float64 (with float64
# logits) to evaluate the same formulas on the same float32 table values without float32 rounding, which tells
# how far the reference's own float32 results are from the exact value -- the floor below which two correct
# float32 evaluations cannot be expected to agree.
ARITH = torch.float32


class arithmetic:
    """``with arithmetic(torch.float64): ...`` -- evaluate in that type inside the block."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global ARITH
        self.saved, ARITH = ARITH, self.dtype

    def __exit__(self, *exc):
        global ARITH
        ARITH = self.saved
        return False


def rows_at(a: torch.Tensor, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``_at`` (:205-226): f32 cast of the whole table, then a[t_b, x]."""
    a = a.to(dtype=torch.float32).to(ARITH)
    tb = t.unsqueeze(1).unsqueeze(-1).unsqueeze(-1)
    return a[tb, x]


def contract_at(a: torch.Tensor, t: torch.Tensor, probs: torch.Tensor) -> torch.Tensor:
    """``_at_onehot`` (:229-247): probs @ a[t_b] as a broadcast matmul."""
    a = a.to(dtype=torch.float32).to(ARITH)
    return torch.matmul(probs, a[t, None, None, Ellipsis])


def gumbel_from_uniform(u: torch.Tensor) -> torch.Tensor:
    """:283-284 / :396-397."""
    u = torch.clip(u, min=torch.finfo(u.dtype).tiny, max=1.0)
    return -torch.log(-torch.log(u))


# --------------------------------------------------------------------------
# forward process
# --------------------------------------------------------------------------
def forward_probs(p: Process, x0: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """``q_probs`` (:250-263)."""
    return rows_at(p.q_mats, t, x0)


def forward_sample(p: Process, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """``q_sample`` (:266-285)."""
    assert noise.shape == x0.shape + (p.K,)
    logits = torch.log(forward_probs(p, x0, t) + p.eps)
    return torch.argmax(logits + gumbel_from_uniform(noise), dim=-1)


# --------------------------------------------------------------------------
# posterior / model logits
# --------------------------------------------------------------------------
def posterior_logits(p: Process, x_start: torch.Tensor, x_t: torch.Tensor, t: torch.Tensor,
                     x_start_is_logits: bool) -> torch.Tensor:
    """``q_posterior_logits`` (:312-336)."""
    if x_start_is_logits:
        assert x_start.shape == x_t.shape + (p.K,)
    else:
        assert x_start.shape == x_t.shape
    f1 = rows_at(p.qT_onestep, t, x_t)
    if x_start_is_logits:
        f2 = contract_at(p.q_mats, t - 1, F.softmax(x_start, dim=-1))
        tzero = x_start
    else:
        f2 = rows_at(p.q_mats, t - 1, x_start)
        tzero = torch.log(F.one_hot(x_start.to(torch.int64), num_classes=p.K).to(ARITH) + p.eps)
    out = torch.log(f1 + p.eps) + torch.log(f2 + p.eps)
    tb = t.unsqueeze(1).unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
    return torch.where(tb == 0, tzero, out)


def model_logits(p: Process, pred_x_start_logits: torch.Tensor, x_t: torch.Tensor, t: torch.Tensor):
    """``p_logits`` (:339-381) for model_output='logits', model_prediction=
    'x_start', with the denoiser output passed in instead of a callback."""
    assert t.shape == (x_t.shape[0],)
    tb = t.unsqueeze(1).unsqueeze(-1).unsqueeze(-1).unsqueeze(-1)
    ml = torch.where(tb == 0, pred_x_start_logits,
                     posterior_logits(p, pred_x_start_logits, x_t, t, True))
    return ml, pred_x_start_logits


def reverse_step(p: Process, pred_x_start_logits: torch.Tensor, x_t: torch.Tensor, t: torch.Tensor,
                 noise: torch.Tensor):
    """``p_sample`` (:384-403)."""
    ml, px0 = model_logits(p, pred_x_start_logits, x_t, t)
    assert noise.shape == ml.shape
    mask = (t != 0).type(x_t.dtype).view(x_t.shape[0], *([1] * (len(x_t.shape))))
    g = gumbel_from_uniform(noise)
    sample = torch.argmax(ml + mask * g, dim=-1)
    return sample, F.softmax(px0, dim=-1)


# --------------------------------------------------------------------------
# d3pm/utils.py helpers
# --------------------------------------------------------------------------
def kl_logits(l1: torch.Tensor, l2: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """utils.py:41-58."""
    out = F.softmax(l1 + eps, dim=-1) * (F.log_softmax(l1 + eps, dim=-1) - F.log_softmax(l2 + eps, dim=-1))
    return torch.sum(out, dim=-1)


def kl_probs(p1: torch.Tensor, p2: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """utils.py:61-75."""
    return torch.sum(p1 * (torch.log(p1 + eps) - torch.log(p2 + eps)), dim=-1)


def log_likelihood(x: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """utils.py:78-92."""
    lp = F.log_softmax(logits, dim=-1)
    oh = F.one_hot(x.to(torch.int64), num_classes=logits.size(-1))
    return torch.sum(lp * oh, dim=-1)


def mean_nonbatch(x: torch.Tensor) -> torch.Tensor:
    """utils.py:95-97."""
    return x.mean(dim=tuple(range(1, x.dim())))


def log_min_exp(a: torch.Tensor, b: torch.Tensor, eps: float = EPS) -> torch.Tensor:
    """utils.py:15-18."""
    return a + torch.log1p(-torch.exp(b - a) + eps)


def logits_from_logistic_pars(loc: torch.Tensor, log_scale: torch.Tensor, K: int, eps: float = EPS) -> torch.Tensor:
    """``_get_logits_from_logistic_pars`` (:288-309): bin logits of a logistic
    with location ``loc`` and scale ``exp(log_scale - 2)`` over K bins centred on
    linspace(-1, 1, K)."""
    loc, log_scale = loc.unsqueeze(-1), log_scale.unsqueeze(-1)
    inv_scale = torch.exp(-(log_scale - 2.0))
    width = 2.0 / (K - 1.0)
    centers = torch.linspace(start=-1.0, end=1.0, steps=K)
    for _ in range(loc.ndim - 1):
        centers = centers.unsqueeze(0)
    centers = centers - loc
    lo = torch.nn.functional.logsigmoid(inv_scale * (centers - 0.5 * width))
    hi = torch.nn.functional.logsigmoid(inv_scale * (centers + 0.5 * width))
    return log_min_exp(hi, lo, eps)


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def vb_terms(p: Process, pred_x_start_logits: torch.Tensor, x0: torch.Tensor, x_t: torch.Tensor,
             t: torch.Tensor):
    """``vb_terms_bpd`` (:462-489)."""
    true_l = posterior_logits(p, x0, x_t, t, False)
    ml, px0 = model_logits(p, pred_x_start_logits, x_t, t)
    kl = mean_nonbatch(kl_logits(true_l, ml)) / math.log(2.0)
    nll = mean_nonbatch(-log_likelihood(x0, ml)) / math.log(2.0)
    return torch.where(t == 0, nll, kl), px0


def ce_start(x0: torch.Tensor, pred_x_start_logits: torch.Tensor) -> torch.Tensor:
    """``cross_entropy_start`` (:527-542)."""
    return mean_nonbatch(-log_likelihood(x0, pred_x_start_logits)) / math.log(2.0)


def prior_term(p: Process, x0: torch.Tensor) -> torch.Tensor:
    """``prior_bpd`` (:492-524)."""
    qp = forward_probs(p, x0, torch.full((x0.shape[0],), p.T - 1))
    if p.kind in ("gaussian", "uniform"):
        prior = torch.ones_like(qp) / p.K
    elif p.kind == "absorbing":
        idx = torch.full(size=qp.shape[:-1], fill_value=p.K // 2, dtype=torch.int64)
        prior = F.one_hot(idx, num_classes=p.K).to(torch.int32)
    else:
        raise ValueError(p.kind)
    return mean_nonbatch(kl_probs(qp, prior)) / math.log(2.0)


def losses_given_xt(p: Process, pred_x_start_logits: torch.Tensor, x0: torch.Tensor,
                    x_t: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """The part of ``training_losses`` after q_sample (:558-582), with the
    denoiser output passed in."""
    if p.loss_type == "kl":
        out, _ = vb_terms(p, pred_x_start_logits, x0, x_t, t)
    elif p.loss_type == "cross_entropy_x_start":
        out = ce_start(x0, pred_x_start_logits)
    elif p.loss_type == "hybrid":
        vb, px0 = vb_terms(p, pred_x_start_logits, x0, x_t, t)
        out = vb + p.hybrid_coeff * ce_start(x0, px0)
    else:
        raise NotImplementedError(p.loss_type)
    assert out.shape == t.shape
    return out


def training_step(p: Process, model_fn: Callable, x0: torch.Tensor, t: torch.Tensor,
                  noise: torch.Tensor):
    """``training_losses`` (:545-582) with t and noise injected: returns
    (losses[B], x_t)."""
    x_t = forward_sample(p, x0, t, noise)
    return losses_given_xt(p, model_fn(x_t, t), x0, x_t, t), x_t


def bpd_sweep(p: Process, model_fn: Callable, x0: torch.Tensor, noises) -> dict:
    """``calc_bpd_loop`` (:584-629): the variational-bound term of every timestep
    (x_t drawn from q with ``noises[t]``, the uniform tensor the reference draws at
    :611) plus the prior term.  Returns total [B], vbterms [B, T], prior [B]."""
    B = x0.shape[0]
    vb_tb = torch.empty((p.T, B), dtype=torch.float32)
    with torch.no_grad():
        for step in range(p.T):
            t = torch.full((B,), step, dtype=torch.int32)
            x_t = forward_sample(p, x0, t, noises[step])
            vb_tb[step] = vb_terms(p, model_fn(x_t, t), x0, x_t, t)[0]
        prior = prior_term(p, x0)
    return {"total": vb_tb.sum(dim=0) + prior, "vbterms": vb_tb.T, "prior": prior}
