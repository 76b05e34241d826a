#!/usr/bin/env python3
"""Data-parallel training harness around the D3PM core: a ByteNet-like denoiser in
stock PyTorch, this package for everything between data and logits gradient,
DistributedDataParallel for the denoiser's gradients (the core itself has no
parameters and no collective).

--style protein (default) is the reference's protein training loop (Protein
Generation Notebook cell-5 / cell-12): the lean Collater corrupts the batch on the
device (t ~ randint(1, T) per sequence, x_t ~ q(x_t | x_0)), the denoiser maps
[B, L] tokens to [B, L, n_tokens] logits, and the loss is
(D3PMLVBLoss + lambda * CrossEntropyLoss) * n_tokens from one fused launch, K = 22,
L = 263, 256 sequences per GPU.  --style image is the d3pm-style two-launch
DiscreteDiffusion.training_step on [B, L, 1, 1] tokens.

    python tools/train_ddp.py --steps 30                              # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29511 tools/train_ddp.py --steps 30               # N GPUs, RCCL

Prints one JSON line (rank 0): step time, the share of the step spent in the path
(corruption + fused loss, measured with HIP events on the launch stream), and the
share added by the gradient all-reduce (step time with and without `no_sync`).
Per-rank normalisation is the reference's: every rank averages over its local batch
and DDP averages the gradients, which reproduces the global-batch mean.
"""
from __future__ import annotations

import argparse
import contextlib
import faulthandler
import json
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


class Block(nn.Module):
    def __init__(self, d, dilation):
        super().__init__()
        self.n1, self.n2 = nn.LayerNorm(d), nn.LayerNorm(d // 2)
        self.down, self.up = nn.Linear(d, d // 2), nn.Linear(d // 2, d)
        self.conv = nn.Conv1d(d // 2, d // 2, 5, padding=2 * dilation, dilation=dilation)

    def forward(self, x):                               # [B, L, d]
        h = self.down(torch.nn.functional.gelu(self.n1(x)))
        h = self.conv(torch.nn.functional.gelu(self.n2(h)).transpose(1, 2)).transpose(1, 2)
        return x + self.up(torch.nn.functional.gelu(h))


class ByteNetLike(nn.Module):
    """Token + timestep embedding, dilated residual conv blocks, linear head."""

    def __init__(self, K, T, d=256, layers=8):
        super().__init__()
        self.tok, self.time = nn.Embedding(K, d), nn.Embedding(T + 1, d)
        self.blocks = nn.ModuleList(Block(d, 2 ** (i % 5)) for i in range(layers))
        self.norm, self.head = nn.LayerNorm(d), nn.Linear(d, K)

    def forward(self, x, t):                            # x [B, L, 1, 1] int64, t [B]
        shape = x.shape
        h = self.tok(x.reshape(shape[0], -1)) + self.time(t.long())[:, None, :]
        for b in self.blocks:
            h = b(h)
        return self.head(self.norm(h)).reshape(*shape, -1).float()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=263)
    ap.add_argument("--style", default="protein", choices=["protein", "image"])
    ap.add_argument("--lam", type=float, default=0.001, help="weight of the cross-entropy term (both styles)")
    ap.add_argument("--bits", type=int, default=5, help="--style image: K = 2**bits classes")
    ap.add_argument("--dump", default=None,
                    help="--style protein: save the last timed step's x0, x_t, t, logits, loss and dlogits of rank 0 "
                         "to this .pt file")
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--backend", default="nccl",
                    help="nccl (= RCCL over xGMI, one rank per GPU) or gloo (rehearsal of the N > 1 code path on a box "
                         "with fewer GPUs than ranks: gradients are staged through the host)")
    ap.add_argument("--bf16", action="store_true", help="autocast the denoiser to bfloat16 (the path stays float32)")
    ap.add_argument("--watchdog", type=float, default=120.0,
                    help="seconds without reaching the next phase after which every thread's stack is dumped and the "
                         "process exits (0 = off)")
    args = ap.parse_args(argv)

    try:
        return run(args)
    finally:
        # armed by every phase() below: must not outlive this call (the caller may be a test process)
        faulthandler.cancel_dump_traceback_later()


def run(args):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    def phase(name):
        """One line per phase on stderr and a fresh watchdog: a stalled collective names itself."""
        print(f"[train_ddp rank {rank}] {name}", file=sys.stderr, flush=True)
        if args.watchdog > 0:
            faulthandler.dump_traceback_later(args.watchdog, exit=True)

    phase("start")
    ngpu = torch.cuda.device_count()
    if ngpu == 0:
        raise SystemExit("train_ddp.py needs a GPU: the D3PM kernels have no CPU path")
    if world > 1 and args.backend == "nccl" and world > ngpu:
        raise SystemExit(f"--backend nccl needs one GPU per rank ({world} ranks, {ngpu} GPU(s)): RCCL does not form a "
                         "communicator with two ranks on one device; use --backend gloo to rehearse the code path")
    dev = torch.device("cuda", local % ngpu)
    torch.cuda.set_device(dev)
    gloo = args.backend == "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        phase(f"init_process_group({args.backend})")
        dist.init_process_group(args.backend, **({} if gloo else {"device_id": dev}))
    # timing / logging collectives: host tensors under gloo (no device staging inside the collective), device
    # tensors under RCCL
    cdev = torch.device("cpu") if gloo else dev

    import d3pm_amd
    protein = args.style == "protein"
    K, T, B, L = (22 if protein else 2 ** args.bits), 500, args.batch_per_gpu, args.seq_len
    torch.manual_seed(1234)                                   # same initial weights on every rank
    model = ByteNetLike(K, T, args.d_model, args.layers).to(dev)
    torch.cuda.synchronize()
    # RCCL: stock DistributedDataParallel (bucketed all-reduce overlapped with backward; identical seeds give
    # identical weights, so the constructor's parameter broadcast is skipped).  gloo: the reducer's all-reduce of
    # DEVICE buckets from inside backward never completes on this ROCm build (both ranks stall in
    # _engine_run_backward), so the rehearsal averages the gradients through host tensors after backward instead.
    use_ddp = world > 1 and not gloo
    phase("wrap denoiser in DistributedDataParallel" if use_ddp else "denoiser built")
    net = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], init_sync=False) if use_ddp else model

    def average_grads_via_host():
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        flat = torch.cat([g_.reshape(-1) for g_ in grads]).cpu()
        dist.all_reduce(flat)
        flat = (flat / world).to(dev)
        off = 0
        for g_ in grads:
            g_.copy_(flat[off:off + g_.numel()].view_as(g_))
            off += g_.numel()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    torch.manual_seed(99 + rank)                              # different data and noise per rank
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def model_fn(x, t):
        ev[1].record()                                        # corruption done
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            out = net(x, t)
        ev[2].record()                                        # logits ready: the loss follows
        return out

    if protein:
        import numpy as np
        from d3pm_amd.protein_generation.dataset import Collater
        np.random.seed(99 + rank)                             # the collater draws its timesteps from numpy's generator
        tok = d3pm_amd.Tokenizer()
        Qb, Qt = tok.q_uniform_schedule(num_steps=T, device=dev)
        collater = Collater(tokenizer=tok, num_steps=T, Q=Qt, Q_bar=Qb, device=dev, lean=True)
        loss_fn = d3pm_amd.D3PMLVBLoss(tmax=T, tokenizer=tok, hybrid_coeff=args.lam)
        data = torch.randint(0, K, (B, L), device=dev)        # tokenised sequences, resident on the device
        last = {}

        def forward_backward():
            src, timesteps, tgt = collater.collate_tokens(data)            # dataset.py:219-250, one launch
            outputs = model_fn(src, timesteps)                              # [B, L, n_tokens]
            n_tokens = src.shape[0] * src.shape[1]                          # notebook cell-12
            if args.dump:
                outputs.retain_grad()
            loss = loss_fn(None, None, outputs, tgt, None, timesteps, Qt, Qb, src=src, out_scale=n_tokens)
            loss.backward()
            last.update(x0=tgt, x_t=src, t=timesteps, logits=outputs, loss=loss.detach())
            # per-sequence lvb + lambda * ce (not scaled by n_tokens), for the log line
            return {"losses": (loss.detach() / n_tokens).reshape(1), "x_t": src, "t": timesteps}
    else:
        d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0.0, 0.0, T, "jsd"),
                                       transition_mat_type="uniform", num_bits=args.bits, transition_bands=None,
                                       model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                       hybrid_coeff=args.lam, device=dev)
        data = torch.randint(0, K, (B, L, 1, 1), device=dev)

        def forward_backward():
            return d.training_step(model_fn, x_start=data)    # backward through the denoiser included

    def step(sync=True):
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        ctx = net.no_sync() if (use_ddp and not sync) else contextlib.nullcontext()
        with ctx:
            out = forward_backward()
        if world > 1 and gloo and sync:
            average_grads_via_host()
        opt.step()
        return out

    def timed(n, sync=True):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = step(sync)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / n * 1e3, out

    phase(f"{args.warmup} warm-up steps")
    first = None
    for _ in range(args.warmup):
        o = step()
        if first is None:
            first = float(o["losses"].double().mean().item())
    phase(f"{args.steps} timed steps")
    ms, out = timed(args.steps)
    # weights stay identical across ranks only if every rank applied the same averaged gradient (checked here,
    # before the steps without all-reduce let the ranks drift apart on purpose)
    wsum = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().to(cdev)
    # share of the path: corruption (ev0 -> ev1) measured inside a step; the fused loss + mean by direct timing
    torch.cuda.synchronize()
    corrupt_ms = ev[0].elapsed_time(ev[1])
    phase("path share")
    with torch.no_grad():
        logits = model(out["x_t"], out["t"]).detach()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if protein:
        if args.dump and rank == 0:
            torch.save({k: v.detach().cpu() for k, v in last.items()} | {"dlogits": last["logits"].grad.detach().cpu(),
                       "lam": args.lam}, args.dump)
        lg = logits.requires_grad_(True)
        fused = lambda: loss_fn(None, None, lg, data, None, out["t"], Qt, Qb, src=out["x_t"], out_scale=B * L)
    else:
        frozen = lambda x, t: logits
        d.training_step(frozen, x_start=data, backward=False)
        fused = lambda: d._fused(logits.requires_grad_(True), data, out["x_t"], out["t"], "hybrid")
    fused()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        fused()
    e1.record(); torch.cuda.synchronize()
    loss_ms = e0.elapsed_time(e1) / 20
    if world > 1:
        phase(f"{args.steps} timed steps without gradient all-reduce")
    nosync_ms = timed(args.steps, sync=False)[0] if world > 1 else ms
    losses = out["losses"].double().mean().to(cdev)
    spread = torch.zeros((), dtype=torch.float64, device=cdev)
    if world > 1:
        phase("final reductions")
        dist.all_reduce(losses)
        losses /= world
        lo, hi = wsum.clone(), wsum.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = (hi - lo).abs() / hi.abs().clamp_min(1e-30)
    result = {
        "what": ("DDP training step, protein style: lean Collater (device corruption) + ByteNet-like denoiser (stock "
                 "PyTorch) + (D3PMLVBLoss + lambda * CE) * n_tokens from one fused launch" if protein else
                 "DDP training step, image style: ByteNet-like denoiser (stock PyTorch) + DiscreteDiffusion.training_step"),
        "style": args.style, "lambda": args.lam,
        "n_gpus": world, "backend": args.backend if world > 1 else None, "global_batch": B * world, "seq_len": L,
        "K": K, "denoiser": {"d_model": args.d_model, "layers": args.layers, "bf16": args.bf16,
                             "params": sum(p.numel() for p in model.parameters())},
        "ms_per_step": ms, "tokens_per_s": B * L * world / (ms * 1e-3),
        "path_ms": {"corrupt_and_timestep_draw": corrupt_ms, "fused_loss_fwd_bwd": loss_ms},
        "path_share": (corrupt_ms + loss_ms) / ms,
        "allreduce_share": max(0.0, (ms - nosync_ms) / ms) if world > 1 else 0.0,
        "first_step_loss": first, "mean_loss": float(losses.item()),
        "loss_unit": "lvb + lambda * ce, nats per sequence position" if protein else "bits per dimension",
        "weight_checksum_spread_over_ranks": float(spread.item())}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    phase_done = f"[train_ddp rank {rank}] done"
    print(phase_done, file=sys.stderr, flush=True)
    return result


if __name__ == "__main__":
    main()
