"""HIP-graph capture of the loss step.

One training step of the image-style process is a handful of short kernels
(timestep draw, forward corruption, fused loss, the caller's reduction and its
backward); launched eagerly from Python the step is bound by host launch
overhead, not by the GPU.  ``GraphedTrainingLosses`` captures the whole step,
including ``loss.backward()``, into one HIP graph and replays it.  Fresh noise
on every replay comes from a device-side Philox counter that the graph itself
advances.
"""
from __future__ import annotations

import torch


class GraphedTrainingLosses:
    """``DiscreteDiffusion.training_losses(model_fn, x_start=...)`` followed by
    ``.mean().backward()``, captured once and replayed.

    ``x_start`` is a static input buffer (copy new data into it between
    replays); ``model_fn`` must be capturable (static shapes, no host syncs);
    gradients land in the ``.grad`` of the tensors ``model_fn`` depends on.
    ``params`` lists those leaf tensors: their ``.grad`` is cleared before the
    capture so that the captured backward writes gradients instead of
    accumulating into the warm-up's (one elementwise pass less per step); read
    ``p.grad`` after ``replay()``.  ``steps`` > 1 captures that many consecutive
    steps in one graph (one host-side graph launch for all of them; ``params``'
    gradients are cleared before each captured step, so after a replay they hold
    the last step's gradient).
    """

    def __init__(self, diffusion, model_fn, x_start: torch.Tensor, seed: int = 0, warmup: int = 3, params=(),
                 steps: int = 1):
        self.diffusion, self.model_fn, self.x_start = diffusion, model_fn, x_start
        self.seed = int(seed)
        self.steps = int(steps)
        dev = x_start.device
        # parameters usually predate the capture stream; the mismatch torch warns about is intended here
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        # builds the digests now (set-up syncs are not capturable) and keeps them: the captured launches hold raw
        # pointers into these buffers, which a later change of q_mats / force_dense would otherwise free
        self._tables = diffusion.tables()
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body().backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        for p in params:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for i in range(self.steps):
                if i:
                    for p in params:
                        p.grad = None
                self.loss = self._body()
                self.loss.backward()

    def _body(self):
        # (the loss launch advances the Philox counter: no bookkeeping launch in the graph)
        losses = self.diffusion.training_losses(self.model_fn, x_start=self.x_start, rng=(self.seed, self.counter))
        self.losses = losses
        return losses.mean()

    def replay(self) -> torch.Tensor:
        """One step; returns the (static) scalar loss tensor."""
        self.graph.replay()
        return self.loss


class GraphedTrainingStep:
    """``DiscreteDiffusion.training_step(model_fn, x_start=...)`` -- timestep draw +
    corruption, the denoiser, the fused loss with batch mean and gradient, and the
    backward through the denoiser -- captured once and replayed.  The Philox
    counter lives on the device and is advanced by the loss kernel, so the graph
    holds no bookkeeping launches at all.  Arguments as for
    ``GraphedTrainingLosses``; ``replay()`` returns the static result dict.

    ``steps`` > 1 captures that many consecutive steps in one graph (a replay then
    costs one host-side graph launch, ~19 us, for all of them: with one step of
    ~16 us of GPU work per graph the host is the bound).  ``x_start`` is then
    either one batch buffer read by every step or a sequence of ``steps`` buffers
    (a ring of pre-staged batches); ``params``' gradients are cleared before each
    captured step, so after a replay they hold the last step's gradient --
    capture the optimiser inside ``model_fn``'s backward hooks or use ``steps=1``
    when every step's gradient is consumed on the host side.  ``outs`` lists the
    result dict of every captured step, ``out`` is the last.

    ``pipelined`` (with ``steps`` > 1): the loss launch of step i also draws the
    timesteps and noisy tokens of step i + 1 (``loss_step(prefetch=...)``), so a graph
    of n steps holds n + 1 launches of the path instead of 2 n; timesteps, tokens,
    losses and gradients are those of the unpipelined graph with the same seed, bit
    for bit."""

    def __init__(self, diffusion, model_fn, x_start, seed: int = 0, warmup: int = 3, params=(), steps: int = 1,
                 pipelined: bool = False):
        self.diffusion, self.model_fn = diffusion, model_fn
        self.batches = list(x_start) if isinstance(x_start, (list, tuple)) else [x_start] * steps
        assert len(self.batches) == steps >= 1
        self.x_start = self.batches[0]
        self.seed, self.steps = int(seed), steps
        self.pipelined = bool(pipelined) and steps > 1
        dev = self.x_start.device
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        self._tables = diffusion.tables()                    # pinned: the graph holds raw pointers into the digests
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if self.pipelined:
                    self._pipeline(params)
                else:
                    self._body(self.x_start)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.outs = []
        with torch.cuda.graph(self.graph):
            if self.pipelined:
                self.outs = self._pipeline(params)
            else:
                for x in self.batches:
                    for p in params:
                        p.grad = None
                    self.outs.append(self._body(x))
        self.out = self.outs[-1]

    def _body(self, x):
        return self.diffusion.training_step(self.model_fn, x_start=x, rng=(self.seed, self.counter))

    def _pipeline(self, params):
        """n steps as one corruption launch and n loss launches: step i's loss launch corrupts batch i + 1 at stream
        offset 4 (i + 1) + counter (what step i + 1 of the unpipelined sequence reads after i + 1 advances of 4);
        the last loss launch advances the counter by 4 n."""
        d, n = self.diffusion, self.steps
        corrupted = d.corrupt_step(x_start=self.batches[0], rng=(self.seed, self.counter))
        outs = []
        for i, x in enumerate(self.batches):
            for p in params:
                p.grad = None
            last = i == n - 1
            out = d.loss_step(self.model_fn, x_start=x, corrupted=corrupted, advance=4 * n if last else 0,
                              prefetch=None if last else {"x_start": self.batches[i + 1], "offset": 4 * (i + 1)})
            corrupted = out.pop("next", None)
            outs.append(out)
        return outs

    def replay(self):
        self.graph.replay()
        return self.out


class GraphedReverseSampler:
    """The T-1 reverse steps of ``generate_d3pm`` (denoiser call + posterior
    marginal + categorical draw per step) for a fixed batch shape, captured into one
    HIP graph.  Launched step by step from Python the loop is bound by the host
    (a few launches of a few microseconds each per step); the replay is bound by
    the kernels.  Every replay draws fresh noise: the Philox offset of each step is
    its capture-time offset plus a device counter that ``sample()`` advances.

    ``model_fn(x [B,L] int64, t [B] int64) -> logits [B,L,n_tokens]`` must be
    capturable.  ``sample(x_init=None)`` returns the static [B,L] result tensor
    (clone it before the next call); x_T is drawn uniformly unless given.
    ``capture=False`` keeps the same object and noise sequence but launches the
    steps one by one (for runs under a kernel tracer, where single launches are
    wanted)."""

    def __init__(self, tables, model_fn, batch_size: int, seq_len: int, timesteps: int, device, seed: int = 0,
                 warmup: int = 2, capture: bool = True):
        from .protein_generation.generate import reverse_step
        self.tables, self.K = tables, tables.K
        dev = torch.device(device)
        self.x = torch.zeros((batch_size, seq_len), dtype=torch.long, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        steps = list(range(timesteps - 1, 0, -1))
        # inputs of the captured launches: they must stay allocated as long as the graph is replayed
        self.tts = tts = [torch.full((batch_size,), t, dtype=torch.long, device=dev) for t in steps]
        self.model_fn = model_fn
        self.stride = 4 * len(steps)

        def body():
            s = self.x
            for i, tt in enumerate(tts):
                s = reverse_step(tables, model_fn(s, tt), s, tt, rng=(seed, 4 * i, self.counter))
            return s

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                reverse_step(tables, model_fn(self.x, tts[0]), self.x, tts[0], rng=(seed, 0, self.counter))
        torch.cuda.current_stream(dev).wait_stream(side)
        self._body = body
        self.graph = None
        if capture:
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = body()

    def sample(self, x_init=None):
        if x_init is None:
            self.x.random_(0, self.K)
        else:
            self.x.copy_(x_init)
        self.counter.add_(self.stride)
        if self.graph is not None:
            self.graph.replay()
        else:
            with torch.no_grad():
                self.out = self._body()
        return self.out


class GraphedBpdSweep:
    """``DiscreteDiffusion.calc_bpd_loop`` -- corruption, denoiser and bound term at
    every timestep, then the prior term and the total -- for a fixed ``x_start``
    buffer, captured into one HIP graph.  Eagerly the sweep is T rounds of a few
    short launches each and is bound by the host; the replay runs them back to
    back.  Every replay corrupts with fresh noise (device Philox counter, advanced
    by 4 T per call).  ``replay()`` returns the static result dict of
    ``calc_bpd_loop``; copy new data into ``x_start`` between calls."""

    def __init__(self, diffusion, model_fn, x_start: torch.Tensor, seed: int = 0):
        self.diffusion, self.model_fn, self.x_start = diffusion, model_fn, x_start
        self.seed = int(seed)
        dev = x_start.device
        self._tables = diffusion.tables()                    # set-up syncs are not capturable; pinned for the graph
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.stride = 4 * diffusion.num_timesteps
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():       # one untimed step of each kind outside the capture
            t0 = torch.zeros(x_start.shape[0], dtype=torch.int32, device=dev)
            x_t = diffusion.q_sample(x_start, t0, rng=(self.seed, self.counter))
            diffusion.vb_terms_bpd(model_fn, x_start=x_start, x_t=x_t, t=t0)
            diffusion.prior_bpd(x_start)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = diffusion.calc_bpd_loop(model_fn, x_start=x_start, rng=(self.seed, self.counter))

    def replay(self) -> dict:
        self.counter.add_(self.stride)
        self.graph.replay()
        return self.out


class GraphedPSampleLoop:
    """``DiscreteDiffusion.p_sample_loop`` (:405-458) for a fixed shape -- the
    denoiser call and the reverse step of every timestep -- captured into one HIP
    graph.  ``sample()`` draws x_T (uniform tokens, or K // 2 for absorbing
    transitions) unless ``x_init`` is given, advances the device Philox counter so
    that every replay draws fresh noise, replays, and returns the static result
    tensor (clone it before the next call).  ``model_fn`` must be capturable."""

    def __init__(self, diffusion, model_fn, shape, seed: int = 0, num_timesteps=None):
        self.diffusion, self.model_fn, self.shape = diffusion, model_fn, tuple(shape)
        dev = torch.device(diffusion.device)
        self._tables = diffusion.tables()                    # set-up syncs are not capturable; pinned for the graph
        self.steps = diffusion.num_timesteps if num_timesteps is None else num_timesteps
        self.x = torch.zeros(self.shape, dtype=torch.int64, device=dev)
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed = int(seed)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                        # one untimed step outside the capture
            diffusion.p_sample_loop(model_fn, shape=self.shape, num_timesteps=1, x_init=self.x,
                                    rng=(self.seed, self.counter))
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = diffusion.p_sample_loop(model_fn, shape=self.shape, num_timesteps=self.steps, x_init=self.x,
                                               rng=(self.seed, self.counter))

    def sample(self, x_init=None):
        if x_init is not None:
            self.x.copy_(x_init)
        elif self.diffusion.transition_mat_type == "absorbing":
            self.x.fill_(self.diffusion.num_pixel_vals // 2)
        else:
            self.x.random_(0, self.diffusion.num_pixel_vals)
        self.counter.add_(4 * self.steps)
        self.graph.replay()
        return self.out
