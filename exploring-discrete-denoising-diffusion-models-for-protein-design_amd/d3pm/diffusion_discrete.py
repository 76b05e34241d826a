"""``DiscreteDiffusion`` with the reference's constructor and method signatures
(d3pm/diffusion_discrete.py:29-629), executed by the HIP kernels of
libd3pm_hip.so.

Differences a caller can see, all additive:
  * tensors must live on a HIP device; there is no CPU execution path;
  * ``q_sample`` / ``p_sample`` accept ``noise_kind`` ('uniform' as in the
    reference, or 'gumbel' for pre-transformed noise) and ``noise=None`` for
    on-device Philox noise; ``training_losses`` accepts optional ``t`` and
    ``noise`` so a run can be replayed;
  * data of any shape [B, ...] is accepted (the reference hard-codes 4-D data).
The loss methods are differentiable with respect to the denoiser output only;
the gradient comes out of the same kernel launch as the loss.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib
from .._ops import ops, i64
from ..tables import ImageTables
from . import utils


def generate_betas(start: float, stop: float, num_steps: int, type: str):
    """Noise schedules (d3pm/diffusion_discrete.py:12-26); float64 tensor [T]."""
    if type == "linear":
        b = np.linspace(start, stop, num_steps)
    elif type == "cosine":
        grid = np.arange(num_steps + 1, dtype=np.float64) / num_steps
        abar = np.cos((grid + 0.008) / 1.008 * np.pi / 2)
        b = np.minimum(1 - abar[1:] / abar[:-1], 0.999)
    elif type == "jsd":
        b = 1.0 / np.linspace(num_steps, 1.0, num_steps)
    else:
        raise NotImplementedError(type)
    return torch.tensor(b)


def _as_tokens(x: torch.Tensor) -> torch.Tensor:
    if x.dtype not in (torch.int32, torch.int64):
        x = x.to(torch.int64)
    return x.contiguous()


def _philox_ticket(device) -> tuple:
    """(seed, offset) from torch's generator for `device`; advances it, so
    torch.manual_seed() makes runs repeatable."""
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, off


_MEAN_SCALE = {}


def _mean_scale(B: int) -> float:
    """float32(1) / float32(B): the upstream gradient `.mean()` hands every batch element."""
    v = _MEAN_SCALE.get(B)
    if v is None:
        v = _MEAN_SCALE[B] = float(np.float32(1.0) / np.float32(max(B, 1)))
    return v


class StepResult(dict):
    """What ``training_step`` returns.  ``['loss']``, the batch mean, is reduced
    from ``['losses']`` whenever it is read (one small launch on the current
    stream, after a graph replay too) unless the step was asked to produce it on
    the device."""

    def __missing__(self, key):
        if key == "loss":
            return self["losses"].mean()
        raise KeyError(key)


class _LogisticLogits(torch.autograd.Function):
    """Bin logits of the discretised logistic and their gradient to (loc,
    log_scale): one kernel each way."""

    @staticmethod
    def forward(ctx, loc, log_scale, centers, eps):
        loc_c, ls_c = loc.contiguous().float(), log_scale.contiguous().float()
        _lib.require_device(loc_c, ls_c, centers)
        out = ops.logistic_logits(loc_c, ls_c, centers, eps)
        ctx.save_for_backward(loc_c, ls_c, centers)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, grad):
        loc, ls, centers = ctx.saved_tensors
        dloc, dls = ops.logistic_logits_backward(loc, ls, centers, grad.contiguous().float(), ctx.eps)
        return dloc, dls, None, None


def _upstream(g: torch.Tensor) -> torch.Tensor:
    """The upstream gradient of a per-example loss as the rescale kernel takes it: float32, dense -- or ONE element
    when every batch element got the same value (``.mean()`` / ``.sum()`` hand down an expanded scalar: stride 0),
    which rescales the whole gradient by that value and needs no copy of the expanded tensor."""
    if g.dim() == 1 and g.numel() > 1 and g.stride(0) == 0:
        g = g[:1]
    return g.contiguous().float()


class _FusedLoss(torch.autograd.Function):
    """loss[B] = (vb | ce | vb + lambda*ce) of one batch; d loss / d logits is
    produced by the same kernel with `baked` as upstream gradient and rescaled
    in place in backward only if the real upstream differs."""

    @staticmethod
    def forward(ctx, logits, x0, x_t, t, tables, loss_type, lam, baked, counter, advance):
        lg = logits.contiguous()
        if lg.dtype != torch.float32:
            lg = lg.float()
        want_grad = ctx.needs_input_grad[0]
        # (counter: the Philox counter of the corruption launch, advanced by this launch)
        loss, dl = ops.hybrid_loss_step(*tables.op_args(), lg, x0, x_t, t, loss_type, float(lam), float(baked), want_grad,
                                        None, counter, int(advance))
        if not want_grad:
            dl = None
        ctx.dl, ctx.baked, ctx.shape, ctx.dtype = dl, float(baked), logits.shape, logits.dtype
        if want_grad:       # kept for a second backward (retain_graph): the kernel is run again with the new upstream
            ctx.again = (lg, x0, x_t, t, tables, loss_type, float(lam))
        return loss

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dl
        if dl is None:
            return (None,) * 10
        g = _upstream(g)
        if getattr(ctx, "spent", False):
            # the first backward gave its buffer away (rescaled in place): recompute the gradient with this
            # upstream gradient passed to the kernel
            lg, x0, x_t, t, tables, loss_type, lam = ctx.again
            gb = g if g.numel() == x0.shape[0] else g.expand(x0.shape[0]).contiguous()
            dl = ops.hybrid_loss(*tables.op_args(), lg, x0, x_t, t, loss_type, lam, 1.0, gb, True)[3]
        else:
            ctx.spent = True
            ops.scale_rows_(dl, g, ctx.baked)
        out = dl.view(ctx.shape)
        if out.dtype != ctx.dtype:
            out = out.to(ctx.dtype)
        return (out,) + (None,) * 9


class _PosteriorFromLogits(torch.autograd.Function):
    """q_posterior_logits(x_start_logits=True) with its gradient to the x_start logits (what autograd derives
    from d3pm/diffusion_discrete.py:320-336 in the reference).  Column-structured tables: one kernel each way;
    dense tables: the backward contracts with the float32 table copies through a library GEMM."""

    @staticmethod
    def forward(ctx, xs, x_t, t, tables):
        K = tables.K
        B = x_t.shape[0]
        npos = x_t.numel() // max(B, 1)
        lg = xs.contiguous()
        if lg.dtype != torch.float32:
            lg = lg.float()
        out = ops.q_posterior_logits(*tables.op_args(), lg, True, x_t, t)
        ctx.save_for_backward(lg, t)
        ctx.tables, ctx.dims, ctx.dtype = tables, (B, npos), xs.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        lg, t = ctx.saved_tensors
        tables, (B, npos) = ctx.tables, ctx.dims
        K = tables.K
        g = g.contiguous().float()
        if tables.structured:
            gz = ops.q_posterior_logits_backward(*tables.op_args(), lg, t, g)
        else:
            tl = t.long()
            pi = torch.softmax(lg.reshape(B, npos, K), dim=-1)
            Qm = tables.qbar[(tl - 1).clamp(min=0)]                           # [B,K,K] float32 (row t-1; unused at t = 0)
            gf2 = g.reshape(B, npos, K) / (torch.bmm(pi, Qm) + tables.eps)
            gpi = torch.bmm(gf2, Qm.transpose(1, 2))
            gz = pi * (gpi - (pi * gpi).sum(-1, keepdim=True))
            gz = torch.where((tl == 0)[:, None, None], g.reshape(B, npos, K), gz).reshape(lg.shape)
        return (gz if gz.dtype == ctx.dtype else gz.to(ctx.dtype)), None, None, None


class _Handoff(torch.autograd.Function):
    """Feeds a gradient that already exists into autograd: the node's backward
    returns the buffer it was given and drops its own reference, so that a leaf
    can adopt the buffer instead of copying it."""

    @staticmethod
    def forward(ctx, pred, holder):
        ctx.holder = holder
        return torch.empty((), dtype=torch.float32, device=pred.device)

    @staticmethod
    def backward(ctx, g):
        return ctx.holder.pop(), None


class DiscreteDiffusion:
    """Discrete state space diffusion process (d3pm/diffusion_discrete.py:29)."""

    def __init__(self, betas, transition_mat_type, num_bits, transition_bands, model_prediction, model_output,
                 loss_type, hybrid_coeff, device, torch_dtype=torch.float32):
        self.device = torch.device(device)
        self.model_prediction = model_prediction    # 'x_start' | 'xprev'
        self.model_output = model_output            # 'logits' | 'logistic_pars'
        self.loss_type = loss_type                  # 'kl' | 'hybrid' | 'cross_entropy_x_start'
        self.hybrid_coeff = hybrid_coeff
        self.torch_dtype = torch_dtype
        self.num_bits = num_bits
        self.num_pixel_vals = 2 ** num_bits
        self.transition_bands = transition_bands
        self.transition_mat_type = transition_mat_type
        self.eps = 1.0e-6

        if not isinstance(betas, torch.Tensor):
            raise ValueError("expected betas to be a torch Tensor")
        if not ((betas > 0).all() and (betas <= 1).all()):
            raise ValueError("betas must be in (0, 1]")
        if transition_mat_type not in _lib.KIND:
            raise ValueError(f"transition_mat_type must be 'gaussian', 'uniform', 'absorbing', "
                             f"but is {transition_mat_type}")
        if self.device.type != "cuda":
            raise RuntimeError("DiscreteDiffusion runs on HIP kernels only: pass device='cuda' "
                               "(there is no CPU path in this package)")
        if torch_dtype != torch.float32:
            raise NotImplementedError("the kernels compute in float32, like the reference's default torch_dtype")
        self.betas = betas.detach().to(dtype=torch.float64)
        self.num_timesteps = T = self.betas.shape[0]
        K = self.num_pixel_vals
        bands = -1 if transition_bands is None else int(transition_bands)
        onestep, q_mats = ops.build_tables(self.betas.to(self.device).contiguous(), _lib.KIND[transition_mat_type], K, bands)
        # q(x_t | x_start) and the transposed one-step matrices, as in :90,:97
        self.q_mats = q_mats
        self.transpose_q_onestep_mats = onestep.permute(0, 2, 1)
        self._tables: Optional[ImageTables] = None
        self._tables_key = None
        self.force_dense = False
        # how dense tables with 256 classes contract (ImageTables): 'f16x2' = float32 emulated on the float16 matrix
        # instructions, 'f32' = the float32 matrix instructions; set before the first call that uses the tables
        self.dense_matmul = "f16x2"

    # ------------------------------------------------------------------ tables
    def tables(self) -> ImageTables:
        """Digest / float32 copies of the current ``q_mats`` and
        ``transpose_q_onestep_mats`` (rebuilt if a caller replaced them)."""
        a, b = self.q_mats, self.transpose_q_onestep_mats
        key = (a.data_ptr(), a._version, b.data_ptr(), b._version, tuple(b.stride()), self.force_dense, self.dense_matmul)
        if self._tables is None or key != self._tables_key:
            with torch.cuda.device(self.device):
                self._tables = ImageTables(a.to(self.device), b.to(self.device), eps=self.eps,
                                           transposed_onestep=True, force_dense=self.force_dense,
                                           dense_matmul=self.dense_matmul)
            self._tables_key = key
        return self._tables

    def _dims(self, x):
        B = x.shape[0]
        return B, (x.numel() // B if B else 0)

    # ------------------------------------------------------------------ gathers (API compatibility)
    def _at(self, a, t, x):
        """a[t_b, x] as float32 (:205-226).  The kernels fuse this gather; the
        method is kept for callers that use it on their own tables."""
        a32 = a.to(dtype=self.torch_dtype)
        tb = t.long().view(-1, *([1] * (x.dim() - 1)))
        return a32[tb, x.long()]

    def _at_onehot(self, a, t, x):
        """x @ a[t_b] (:229-247), kept for API compatibility."""
        a32 = a.to(dtype=self.torch_dtype)[t.long()]
        return torch.matmul(x, a32.view(a32.shape[0], *([1] * (x.dim() - 2)), *a32.shape[1:]))

    # ------------------------------------------------------------------ forward process
    def q_probs(self, x_start, t):
        """q(x_t | x_start) as float32 [..., K] (:250-263)."""
        x = _as_tokens(x_start)
        tb = self.tables()
        _lib.require_device(x)
        t = _as_tokens(t)
        return ops.q_probs(*tb.op_args(), x, t)

    def q_sample(self, x_start, t, noise=None, noise_kind: str = "uniform", rng=None):
        """x_t ~ q(x_t | x_start) by Gumbel-max (:266-285); int64.
        ``rng=(seed, counter[, offset])`` with ``counter`` a one-element int64
        device tensor selects the Philox stream explicitly: stream offset =
        ``offset`` + the counter's value at execution time (for HIP-graph
        capture, where the counter is advanced between or inside replays)."""
        x = _as_tokens(x_start)
        _lib.require_device(x)
        tb = self.tables()
        B, npos = self._dims(x)
        t = _as_tokens(t)
        seed = off = 0
        off_dev = None
        if noise is None:
            kind = _lib.NOISE_PHILOX
            if rng is not None:
                seed, off_dev = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, rng[1]
                off = int(rng[2]) if len(rng) > 2 else 0
                assert off_dev.dtype == torch.int64 and off_dev.is_cuda
            else:
                seed, off = _philox_ticket(x.device)
        else:
            assert noise.shape == x.shape + (tb.K,), noise.shape
            kind = _lib.NOISE[noise_kind]
            noise = noise.contiguous().float()
        return ops.q_sample(*tb.op_args(), x, t, noise, kind, i64(seed), i64(off), off_dev)

    # ------------------------------------------------------------------ posterior
    def _get_logits_from_logistic_pars(self, loc, log_scale):
        """Logits of a discretised logistic (:288-309, utils.log_min_exp :15-18);
        differentiable with respect to both parameters."""
        assert loc.shape == log_scale.shape, (loc.shape, log_scale.shape)
        centers = getattr(self, "_bin_centers", None)
        if centers is None or centers.device != loc.device:
            # the reference's own expression for the centres (evaluated on the host: one rounding behaviour)
            centers = self._bin_centers = torch.linspace(start=-1.0, end=1.0,
                                                         steps=self.num_pixel_vals).to(loc.device)
        return _LogisticLogits.apply(loc, log_scale, centers, float(self.eps))

    def q_posterior_logits(self, x_start, x_t, t, x_start_logits):
        """Logits of q(x_{t-1} | x_t, x_start) (:312-336)."""
        K = self.num_pixel_vals
        if x_start_logits:
            assert x_start.shape == x_t.shape + (K,), (x_start.shape, x_t.shape)
            xs = x_start.contiguous().float()
        else:
            assert x_start.shape == x_t.shape, (x_start.shape, x_t.shape)
            xs = _as_tokens(x_start)
        xt, t = _as_tokens(x_t), _as_tokens(t)
        _lib.require_device(xs, xt)
        tb = self.tables()
        if x_start_logits and torch.is_grad_enabled() and x_start.requires_grad:
            return _PosteriorFromLogits.apply(x_start, xt, t, tb)
        return ops.q_posterior_logits(*tb.op_args(), xs.detach(), bool(x_start_logits), xt, t)

    def _denoiser_logits(self, model_fn, x, t):
        out = model_fn(x, t)
        if self.model_output == "logits":
            return out
        if self.model_output == "logistic_pars":
            loc, log_scale = out
            return self._get_logits_from_logistic_pars(loc, log_scale)
        raise NotImplementedError(self.model_output)

    def p_logits(self, model_fn, *, x, t):
        """Logits of p(x_{t-1} | x_t) and the predicted x_start logits (:339-381)."""
        assert t.shape == (x.shape[0],)
        pred = self._denoiser_logits(model_fn, x, t)
        if self.model_prediction != "x_start":
            raise NotImplementedError(self.model_prediction)
        ml = self.q_posterior_logits(pred, x, t, x_start_logits=True)      # differentiable with respect to pred
        assert ml.shape == pred.shape == x.shape + (self.num_pixel_vals,)
        return ml, pred

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def p_sample(self, model_fn, *, x, t, noise=None, noise_kind: str = "uniform", rng=None):
        """One reverse step (:384-403): (sample int64, softmax(pred_x_start)).
        ``rng=(seed, counter[, offset])`` as in ``q_sample`` (device noise)."""
        assert t.shape == (x.shape[0],)
        pred = self._denoiser_logits(model_fn, x, t)
        if self.model_prediction != "x_start":
            raise NotImplementedError(self.model_prediction)
        return self._p_sample_from_logits(pred, x, t, noise, noise_kind, want_probs=True, rng=rng)

    def _p_sample_from_logits(self, pred, x, t, noise, noise_kind, want_probs, rng=None):
        K = self.num_pixel_vals
        lg = pred.contiguous().float()
        xt, t = _as_tokens(x), _as_tokens(t)
        _lib.require_device(lg, xt)
        assert lg.shape == xt.shape + (K,), lg.shape
        tb = self.tables()
        B, npos = self._dims(xt)
        seed = off = 0
        off_dev = None
        if noise is None:
            kind = _lib.NOISE_PHILOX
            if rng is not None:
                seed, off_dev = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, rng[1]
                off = int(rng[2]) if len(rng) > 2 else 0
                assert off_dev.dtype == torch.int64 and off_dev.is_cuda
            else:
                seed, off = _philox_ticket(xt.device)
        else:
            assert noise.shape == lg.shape, noise.shape
            kind = _lib.NOISE[noise_kind]
            noise = noise.contiguous().float()
        sample, probs = ops.p_sample(*tb.op_args(), lg, xt, t, noise, kind, i64(seed), i64(off), off_dev,
                                     bool(want_probs))
        return sample, (probs if want_probs else None)

    @torch.no_grad()
    def p_sample_loop(self, model_fn, *, shape, num_timesteps=None, return_x_init=False, return_im_steps=False,
                      save_interval=None, x_init=None, rng=None):
        """Ancestral sampling (:405-458); noise is drawn on device per step.
        ``x_init`` (optional) replaces the draw of x_T; ``rng=(seed, counter)``: step
        i uses Philox offset 4 i + the device counter (``graphs.GraphedPSampleLoop``
        captures the whole loop that way)."""
        shape = tuple(shape)
        K = self.num_pixel_vals
        if x_init is not None:
            assert tuple(x_init.shape) == shape, (x_init.shape, shape)
        elif self.transition_mat_type in ("gaussian", "uniform"):
            x_init = torch.randint(low=0, high=K, size=shape, device=self.device)
        elif self.transition_mat_type == "absorbing":
            x_init = torch.full(size=shape, fill_value=K // 2, dtype=torch.int32, device=self.device)
        else:
            raise ValueError(f"transition_mat_type must be 'gaussian', 'uniform', 'absorbing', "
                             f"but is {self.transition_mat_type}")
        if num_timesteps is None:
            num_timesteps = self.num_timesteps
        x = x_init
        keep = [x.detach().cpu()] if (return_im_steps and save_interval is not None) else None
        for i in range(num_timesteps):
            t = torch.full([shape[0]], self.num_timesteps - 1 - i, device=self.device)
            pred = self._denoiser_logits(model_fn, x, t)
            x, _ = self._p_sample_from_logits(pred, x, t, None, "uniform", want_probs=False,
                                              rng=None if rng is None else (rng[0], rng[1], 4 * i))
            if keep is not None and (i + 1) % save_interval == 0:
                keep.append(x.detach().cpu())
        if keep is not None:
            return keep
        assert x.shape == shape
        return (x_init, x) if return_x_init else x

    # ------------------------------------------------------------------ losses
    def _fused(self, pred, x_start, x_t, t, loss_type: str, baked: Optional[float] = None, counter=None, advance=0):
        x0, xt, t = _as_tokens(x_start), _as_tokens(x_t), _as_tokens(t)
        _lib.require_device(pred, x0, xt)
        assert pred.shape == x0.shape + (self.num_pixel_vals,), pred.shape
        B = x0.shape[0]
        if baked is None:
            baked = _mean_scale(B)                                    # upstream of `.mean()`
        return _FusedLoss.apply(pred, x0, xt, t, self.tables(), _lib.LOSS[loss_type], self.hybrid_coeff or 0.0, baked,
                                counter, advance)

    def vb_terms_bpd(self, model_fn, *, x_start, x_t, t):
        """Variational-bound terms in bits/dim (:462-489): (vb[B], pred logits)."""
        pred = self._denoiser_logits(model_fn, x_t, t)
        if self.model_prediction != "x_start":
            raise NotImplementedError(self.model_prediction)
        vb = self._fused(pred, x_start, x_t, t, "kl")
        assert vb.shape == t.shape == (x_start.shape[0],)
        return vb, pred

    def prior_bpd(self, x_start):
        """KL(q(x_{T-1}|x_start) || stationary) in bits/dim (:492-524)."""
        if self.transition_mat_type not in ("gaussian", "uniform", "absorbing"):
            raise ValueError(f"transition_mat_type must be 'gaussian', 'uniform', 'absorbing', "
                             f"but is {self.transition_mat_type}")
        x = _as_tokens(x_start)
        _lib.require_device(x)
        tb = self.tables()
        B, npos = self._dims(x)
        return ops.prior_bpd(*tb.op_args(), x, self.transition_mat_type == "absorbing")

    def cross_entropy_start(self, x_start, pred_x_start_logits):
        """Cross entropy of x_start under the predicted logits, bits/dim (:527-542)."""
        x0 = _as_tokens(x_start)
        t0 = torch.zeros(x0.shape[0], dtype=torch.int32, device=x0.device)
        ce = self._fused(pred_x_start_logits, x0, x0, t0, "cross_entropy_x_start")
        assert ce.shape == (x_start.shape[0],)
        return ce

    def training_losses(self, model_fn, *, x_start, t=None, noise=None, noise_kind: str = "uniform", rng=None):
        """Per-example training loss (:545-582).  `t` and `noise` default to
        on-device draws (uniform t in [0, T), Philox noise); `rng` as in q_sample."""
        if self.loss_type not in _lib.LOSS:
            raise NotImplementedError(self.loss_type)
        B = x_start.shape[0]
        counter, advance = None, 0
        if t is None and noise is None:
            # timesteps and noisy tokens from one launch on the device Philox stream (:549-556); with an explicit
            # counter the loss launch below advances it, so a captured step needs no bookkeeping launch
            c = self.corrupt_step(x_start=x_start, rng=rng)
            t, x_t = c["t"], c["x_t"]
            if rng is not None and len(rng) == 2:
                counter, advance = rng[1], 4
        else:
            if t is None:
                t = torch.randint(0, self.num_timesteps, (B,), dtype=torch.int32, device=self.device)
            x_t = self.q_sample(x_start=x_start, t=t, noise=noise, noise_kind=noise_kind, rng=rng)
        pred = self._denoiser_logits(model_fn, x_t, t)
        if self.model_prediction != "x_start":
            raise NotImplementedError(self.model_prediction)
        losses = self._fused(pred, x_start, x_t, t, self.loss_type, counter=counter, advance=advance)
        assert losses.shape == t.shape
        return losses

    def training_step(self, model_fn, *, x_start, rng=None, backward=True, device_mean=False):
        """One training step of the path in two launches around the denoiser.

        The reference's step is ``training_losses(...)`` (:545-582) followed by the
        caller's ``.mean().backward()``.  Here the timestep draw and the forward
        corruption are one kernel, and the fused loss kernel also produces the
        batch mean and d mean / d logits, which is handed straight to
        ``torch.autograd.backward(logits, dlogits)`` -- no reduction, fill, divide or
        rescale launches in between.  Noise and timesteps come from the device
        Philox stream: ``rng=(seed, counter)`` with ``counter`` a one-element int64
        device tensor (advanced by the loss kernel itself, for HIP-graph replay),
        or torch's CUDA generator when ``rng`` is None.

        Returns a dict: ``losses`` [B], ``t`` [B] int32, ``x_t``, and ``loss`` (scalar
        float32 tensor, mean over the batch: reduced from ``losses`` when read, or
        with ``device_mean`` accumulated by the loss kernel itself with one atomic
        per batch element, which lengthens that kernel by ~2.8 us at B = 256);
        gradients -- of the batch mean either way -- are accumulated into whatever
        the logits depend on when ``backward`` is true."""
        corrupted = self.corrupt_step(x_start=x_start, rng=rng, device_mean=device_mean)
        return self.loss_step(model_fn, x_start=x_start, corrupted=corrupted, backward=backward)

    def corrupt_step(self, *, x_start, rng=None, offset=0, device_mean=False):
        """First half of ``training_step``: one launch drawing the timesteps and the
        noisy tokens.  ``rng=(seed, counter)``: Philox stream offset = ``offset`` + the
        counter's value when the kernel runs.  It depends on nothing but ``x_start`` and
        the counter.  Both halves go on ONE stream: this launch reads the counter and
        the loss launch advances it, and nothing but stream order stands between that
        read and that write, so ``loss_step`` refuses to run on another stream than this
        call's (to overlap the next batch's corruption with a loss, use
        ``loss_step(prefetch=...)``: one launch does both).  Returns the dict
        ``loss_step`` takes."""
        if self.loss_type not in _lib.LOSS:
            raise NotImplementedError(self.loss_type)
        if self.model_prediction != "x_start":
            raise NotImplementedError(self.model_prediction)
        x0 = _as_tokens(x_start)
        _lib.require_device(x0)
        tb = self.tables()
        B, npos = self._dims(x0)
        dev = x0.device
        seed, off = 0, int(offset)
        counter = None
        if rng is not None:
            seed, counter = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, rng[1]
            assert counter.dtype == torch.int64 and counter.is_cuda and counter.numel() == 1
        else:
            seed, off = _philox_ticket(dev)
        mean = torch.empty((), dtype=torch.float32, device=dev) if device_mean else None   # zeroed by this launch
        t, x_t = ops.q_sample_draw_t(*tb.op_args(), x0, i64(seed), i64(off), counter, mean)
        return {"t": t, "x_t": x_t, "mean": mean, "counter": counter, "seed": seed,
                "stream": torch.cuda.current_stream(dev).cuda_stream}

    def loss_step(self, model_fn, *, x_start, corrupted, backward=True, advance=4, prefetch=None):
        """Second half of ``training_step``: the denoiser, then one launch for the
        per-example losses and d mean / d logits, handed to autograd.  ``advance``:
        what the loss kernel adds to the Philox counter of ``corrupt_step`` (4 = one
        step; a pipelined sequence of n steps passes 0 for all but the last, 4 n there).

        ``prefetch={'x_start': next batch, 'offset': its stream offset}``: the same
        launch also does ``corrupt_step`` for the NEXT batch (same Philox stream, same
        result as ``corrupt_step(x_start=next, rng=..., offset=...)``) while the loss's
        rows are on their way from memory; the result dict then carries it under
        ``'next'``, ready to be the ``corrupted`` of the following ``loss_step``.  The
        counter is only read by such a launch (``advance`` must be 0)."""
        x0 = _as_tokens(x_start)
        tb = self.tables()
        B, npos = self._dims(x0)
        dev = x0.device
        t, x_t, mean, counter = corrupted["t"], corrupted["x_t"], corrupted["mean"], corrupted["counter"]
        if corrupted.get("stream", None) is not None and \
                torch.cuda.current_stream(dev).cuda_stream != corrupted["stream"]:
            raise RuntimeError("loss_step must be issued on the stream corrupt_step ran on (the two launches share a "
                               "Philox counter ordered by one stream only); for overlap use loss_step(prefetch=...)")
        pred = self._denoiser_logits(model_fn, x_t, t)
        assert pred.shape == x0.shape + (tb.K,), pred.shape
        lg = pred.detach().contiguous()
        if lg.dtype != torch.float32:
            lg = lg.float()
        want_grad = backward and pred.requires_grad
        nxt = None
        scale = _mean_scale(B)
        lt, lam = _lib.LOSS[self.loss_type], float(self.hybrid_coeff or 0.0)
        if prefetch is None:
            losses, dl = ops.hybrid_loss_step(*tb.op_args(), lg, x0, x_t, t, lt, lam, scale, want_grad, mean, counter,
                                              int(advance) if counter is not None else 0)
        else:
            assert mean is None and (counter is None or int(advance) == 0), \
                "a launch that corrupts the next batch only reads the counter and keeps no device mean"
            nx0 = _as_tokens(prefetch["x_start"])
            _lib.require_device(nx0)
            assert nx0.shape == x0.shape, (nx0.shape, x0.shape)
            if counter is not None:
                seed, off = int(corrupted["seed"]), int(prefetch.get("offset", 0))
            else:
                seed, off = _philox_ticket(dev)
            losses, dl, nt, nxt_tokens = ops.hybrid_loss_then_q_sample(*tb.op_args(), lg, x0, x_t, t, lt, lam, scale,
                                                                       want_grad, nx0, i64(seed), i64(off), counter)
            nxt = {"t": nt, "x_t": nxt_tokens, "mean": None, "counter": counter, "seed": seed,
                   "stream": torch.cuda.current_stream(dev).cuda_stream}
        if want_grad:
            holder = [dl.view(pred.shape) if dl.dtype == pred.dtype else dl.view(pred.shape).to(pred.dtype)]
            del dl
            torch.autograd.backward(_Handoff.apply(pred, holder), self._unit(dev))
        out = StepResult(losses=losses, t=t, x_t=x_t)
        if mean is not None:
            out["loss"] = mean
        if nxt is not None:
            out["next"] = nxt
        return out

    def _unit(self, dev):
        one = getattr(self, "_one", None)
        if one is None or one.device != dev:
            one = self._one = torch.ones((), dtype=torch.float32, device=dev)
        return one

    @torch.no_grad()
    def calc_bpd_loop(self, model_fn, *, x_start, noise=None, rng=None):
        """Variational bound over all timesteps (:584-629): x_t ~ q(x_t | x_0) and
        the bound's term at every t, plus the prior term.  ``noise`` (optional,
        indexable by t, each ``x_start.shape + (K,)`` uniform) replaces the
        reference's per-step ``torch.rand`` (:611); without it the corruption
        draws Philox noise on the device (``rng=(seed, counter)`` as in q_sample:
        step t uses stream offset 4 t + counter, which makes the whole sweep
        capturable, see ``GraphedBpdSweep``)."""
        B = x_start.shape[0]
        vb_tb = torch.empty((self.num_timesteps, B), dtype=torch.float32, device=self.device)
        ts = torch.arange(self.num_timesteps, dtype=torch.int32, device=self.device)[:, None].expand(-1, B).contiguous()
        for step in range(self.num_timesteps):
            t = ts[step]
            x_t = self.q_sample(x_start=x_start, t=t, noise=None if noise is None else noise[step],
                                rng=None if rng is None else (rng[0], rng[1], 4 * step))
            vb_tb[step], _ = self.vb_terms_bpd(model_fn=model_fn, x_start=x_start, x_t=x_t, t=t)
        prior = self.prior_bpd(x_start=x_start)
        total = vb_tb.sum(dim=0) + prior
        assert prior.shape == total.shape == (B,)
        return {"total": total, "vbterms": vb_tb.T, "prior": prior}
