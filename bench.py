#!/usr/bin/env python3
"""Benchmark of the D3PM hot path on MI355X.

One step = one pass of the hot path over one synthetic batch of the protein
configuration (B=256, L=512, K=32, T=500, uniform transitions): forward
corruption `q_sample` (on-device Philox noise) + the fused posterior / hybrid
loss forward+backward kernel producing d loss / d logits for `.mean()` + the
backward hand-off of the autograd node (a rescale launch that exits early).
The denoiser is excluded: its logits are a pre-filled device tensor.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches one process per GPU (torchrun env); the batch is
sharded with no data-path collective (weak scaling: every rank runs the full
per-GPU batch).  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "D3PM loss fwd+bwd tokens/s & reverse-sample seqs/s, 1/2/4/8 MI355X"
B, L, K, T = 256, 512, 32, 500
LAMBDA = 0.001
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
BYTES_PER_TOKEN = 8 * K + 16   # logits in + dlogits out (float32) + x0 + x_t (int64)
# HBM bytes per launch of the fused loss kernel from the PMC passes kept under profiles/
# (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), or None when no profile is committed.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_bench_pmc_hbm.json")


def measured_traffic():
    try:
        prof = json.load(open(TRAFFIC_FILE))
        for name, ctr in prof.items():
            if "k_hybrid_loss" in name and "true>(" in name[name.index("k_hybrid_loss"):] \
                    and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                return (2 * ctr["FETCH_SIZE"]["mean_per_launch"] + ctr["WRITE_SIZE"]["mean_per_launch"]) * 1024
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(budget_s: float = 12.0):
    """The CPU oracle (a restatement of the reference's eager PyTorch path: float64
    tables cast per call, every [N,K] intermediate materialised, autograd
    backward) timed on this host's cores on the same workload."""
    import torch
    from oracle import image_style as im
    cores = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)
    betas = im.make_betas(0.0, 0.0, T, "jsd")
    p = im.build_process(betas, "uniform", 5, None, hybrid_coeff=LAMBDA)
    x0 = torch.randint(0, K, (B, L, 1, 1), generator=g)
    logits = torch.randn(B, L, 1, 1, K, generator=g, requires_grad=True)

    def step():
        noise = torch.rand(B, L, 1, 1, K)
        t = torch.randint(0, T, (B,), dtype=torch.int32)
        loss, _ = im.training_step(p, lambda x, tt: logits, x0, t, noise)
        logits.grad = None
        loss.mean().backward()

    step()
    n, t0 = 0, time.perf_counter()
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    return {"value": B * L * n / el, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of the same workload (B={B}, L={L}, K={K}, T={T}, uniform, hybrid loss fwd+bwd "
                      f"incl. q_sample and noise draw) in {el:.1f} s, torch {torch.__version__} CPU, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--compat", action="store_true",
                    help="time training_losses(...).mean().backward() instead of the two-launch training_step")
    ap.add_argument("--steps-per-graph", type=int, default=8,
                    help="consecutive steps captured in one HIP graph (a graph launch costs ~19 us of host time: with one "
                         "step per graph the host, not the GPU, bounds the step rate)")
    ap.add_argument("--kind", default="uniform", choices=["uniform", "absorbing"])
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (reverse sampling, K=256 image configuration); for counter passes")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the process group (RCCL) is only needed for N > 1; D3PM_BENCH_FORCE_PG=1 creates it for a single rank too, to
    # rehearse the barrier / all-reduce calls of the multi-GPU run on a one-GPU box (launch through torch.distributed.run)
    use_pg = world > 1 or bool(os.environ.get("D3PM_BENCH_FORCE_PG"))
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (the process group is created after every HIP graph below has been captured: no communicator thread is
    # alive while a stream is capturing)

    import d3pm_amd
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables

    torch.manual_seed(1234 + rank)
    betas = d3pm_amd.generate_betas(0.0, 0.0, T, "jsd")
    d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type=args.kind, num_bits=5, transition_bands=None,
                                   model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                   hybrid_coeff=LAMBDA, device=dev)
    assert d.tables().structured
    x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
    logits = torch.randn(B, L, 1, 1, K, device=dev, requires_grad=True)
    model_fn = lambda x, tt: logits

    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS

    # Default: DiscreteDiffusion.training_step -- timestep draw + q_sample (one launch, device Philox), fused loss
    # per example and d mean / d logits (one launch), gradient handed to autograd -- replayed as a HIP graph.
    # --compat: the reference-shaped calls, training_losses(...).mean().backward(), which add the caller's
    # reduction and its backward (four more short launches on [B] tensors).
    def make_step(compat, graph):
        if compat and not graph:
            def step():
                t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
                x_t = d.q_sample(x0, t)                               # Philox noise on device
                losses = d._fused(logits, x0, x_t, t, "hybrid")      # fused fwd + bwd kernel (C ABI d3pm_hybrid_loss)
                logits.grad = None
                losses.mean().backward()                              # hands dlogits to autograd (early-exit rescale)
            return step
        if compat:
            return d3pm_amd.graphs.GraphedTrainingLosses(d, model_fn, x0, seed=1234 + rank, params=[logits]).replay
        if not graph:
            def step():
                logits.grad = None
                d.training_step(model_fn, x_start=x0)
            return step
        return d3pm_amd.graphs.GraphedTrainingStep(d, model_fn, x0, seed=1234 + rank, params=[logits]).replay

    step = make_step(args.compat, not args.no_graph)
    other = make_step(not args.compat, not args.no_graph) if not args.no_secondary else None
    # n consecutive steps per graph replay (default step shape only): K timed steps = K // n replays of the n-step
    # graph + K % n replays of the one-step graph
    spg = args.steps_per_graph if not (args.compat or args.no_graph) else 1
    multi = (d3pm_amd.graphs.GraphedTrainingStep(d, model_fn, x0, seed=4321 + rank, params=[logits], steps=spg).replay
             if spg > 1 else None)

    def run_steps(k):
        if multi is not None:
            for _ in range(k // spg):
                multi()
            k %= spg
        for _ in range(k):
            step()

    # secondary: reverse sampling of 10k BLAT-length sequences, sharded; the 499-step loop of a sub-batch is one graph
    samplers, sub = None, 0
    if not args.no_secondary:
        n_local = (10000 + world - 1) // world
        sub = min(1024, n_local)
        tok = d3pm_amd.Tokenizer()
        Qb, Qt = tok.q_uniform_schedule(num_steps=500, device=dev)
        ptab = protein_tables(Qt, Qb)
        rem = n_local % sub                        # the shard's last, partial sub-batch gets a graph of its own
        frozen = torch.randn(sub, 263, 22, device=dev)
        emb = torch.randn(22, 64, device=dev) * 0.5
        tvec = torch.randn(501, 64, device=dev) * 0.1
        head = torch.randn(64, 24, device=dev) * 0.3
        tiny = lambda x, tt: (emb[x] + tvec[tt][:, None, :]) @ head
        frozen_fn = lambda x, tt: frozen[:x.shape[0]]
        # [path only, with a tiny denoiser] x [full sub-batch, remainder]
        samplers = [[d3pm_amd.graphs.GraphedReverseSampler(ptab, fn, nb, 263, 500, dev, seed=77 + rank + 1000 * k)
                     for k, nb in enumerate((sub, rem)) if nb > 0] for fn in (frozen_fn, tiny)]
    torch.cuda.synchronize()
    if use_pg:
        dist.init_process_group("nccl", device_id=dev)

    # bring the GPU to its sustained clocks first (a step is ~20 us: a short run would otherwise be timed on a
    # part that has just left idle); then the W warm-up steps and the K timed ones
    ramp_until = time.perf_counter() + 0.2
    while time.perf_counter() < ramp_until:
        run_steps(48)
        torch.cuda.synchronize()
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_pg:
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    tokens = B * L * args.steps * world

    single_ms = None
    if multi is not None:           # for the record: one step per graph replay (host-launch-bound)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - s0) * 5.0
    other_ms = None
    if not args.no_secondary:       # the other call shape, for the record (same graph/eager mode, 100 steps)
        for _ in range(10):
            other()
        torch.cuda.synchronize()
        o0 = time.perf_counter()
        for _ in range(100):
            other()
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - o0) * 10.0

    # ---- duration of the dominant kernel: the same launch (same buffers) back to back on the
    # stream, HIP events at both ends; rocprofv3 --kernel-trace of this command reports the same
    # kernel's average duration (profiles/).
    tb = d.tables()
    t_s = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
    xt_s = d.q_sample(x0, t_s)
    lg_s, dl_s = logits.detach(), torch.empty_like(logits)
    loss_s = torch.empty(B, device=dev)
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(max(wsb // 4, 1), device=dev)
    sp = stream_ptr()

    def fused_only():
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg_s), K, ptr(x0), 8, ptr(xt_s), 8, ptr(t_s), 4, B, L,
                                   LOSS["hybrid"], LAMBDA, 1.0 / B, None, None, None, ptr(loss_s), ptr(dl_s), None, None, 0, ptr(ws),
                                   wsb, sp))

    for _ in range(10):
        fused_only()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        fused_only()
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / args.steps

    done = done_e2e = 0
    rev_el = e2e_el = 1.0
    if not args.no_secondary:
        def run_sampler(smps):
            """This rank's share of the 10 000 sequences through the 499-step loop, one HIP graph replay per
            sub-batch (the last one partial)."""
            for smp in smps:
                smp.sample()
            torch.cuda.synchronize()
            if use_pg:
                dist.barrier()
            r0 = time.perf_counter()
            n = 0
            for _ in range(n_local // sub):
                smps[0].sample()
                n += sub
            if rem:
                smps[1].sample()
                n += rem
            torch.cuda.synchronize()
            el = time.perf_counter() - r0
            if use_pg:
                tt_ = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                el = float(tt_.item())
            return n, el

        done, rev_el = run_sampler(samplers[0])
        done_e2e, e2e_el = run_sampler(samplers[1])

    # ---- secondary: the image configuration (B=128, 32x32x1, K=256, T=1000, Gaussian Q): dense contraction on
    # the matrix cores, fused loss fwd+bwd; priced against the dense float32 MFMA peak
    dense = None
    if rank == 0 and not args.no_secondary:
        KI, TI, BI, LI = 256, 1000, 128, 1024
        di = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(1e-4, 0.02, TI, "linear"),
                                        transition_mat_type="gaussian", num_bits=8, transition_bands=None,
                                        model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                        hybrid_coeff=LAMBDA, device=dev)
        tbi = di.tables()
        x0i = torch.randint(0, KI, (BI, 32, 32, 1), device=dev)
        ti = torch.randint(0, TI, (BI,), dtype=torch.int32, device=dev)
        xti = di.q_sample(x0i, ti)
        lgi = torch.randn(BI, 32, 32, 1, KI, device=dev)
        dli = torch.empty_like(lgi)
        lossi = torch.empty(BI, device=dev)
        wsbi = lib.d3pm_hybrid_loss_workspace(BI, LI, KI)
        wsi = torch.empty(max(wsbi // 4, 1), device=dev)

        def dense_only():
            check(lib.d3pm_hybrid_loss(tbi.desc(), ptr(lgi), KI, ptr(x0i), 8, ptr(xti), 8, ptr(ti), 4, BI, LI,
                                       LOSS["hybrid"], LAMBDA, 1.0 / BI, None, None, None, ptr(lossi), ptr(dli), None, None, 0, ptr(wsi),
                                       wsbi, sp))
        for _ in range(20):
            dense_only()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            dense_only()
        e1.record()
        torch.cuda.synchronize()
        dms = e0.elapsed_time(e1) / 100
        tfl = 4.0 * KI * KI * BI * LI / (dms * 1e-3) / 1e12
        dense = {"workload": "image D3PM B=128 32x32x1 K=256 T=1000 Gaussian Q, fused loss fwd+bwd to dlogits",
                 "kernel_ms": dms, "tokens_per_s": BI * LI / (dms * 1e-3),
                 "roofline": {"bound": "mfma", "achieved": tfl, "peak": 157.3, "unit": "TFLOP/s", "frac": tfl / 157.3,
                              "flops_per_token": 4 * KI * KI, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}}
        del di, tbi, lgi, dli

    # ---- secondary: the protein-style training loss (D3PMLVBLoss forward + gradient, float64 arithmetic) and the
    # batched forward corruption of the collaters, B=256 L=512 K=22 n_tokens=24, through the C ABI
    lvb = None
    if rank == 0 and not args.no_secondary:
        BP, LP, KP, LDP = 256, 512, 22, 24
        x0p = torch.randint(0, KP, (BP, LP), device=dev)
        tp = torch.randint(2, 500, (BP,), device=dev)
        xtp = torch.empty_like(x0p)
        predp = torch.randn(BP, LP, LDP, device=dev)
        dpp = torch.empty_like(predp)
        perp = torch.empty(BP, dtype=torch.float64, device=dev)
        wsbp = lib.d3pm_lvb_loss_workspace(BP, LP, KP)
        wsp = torch.empty(max(wsbp // 8, 1), dtype=torch.float64, device=dev)

        def lvb_step():
            check(lib.d3pm_corrupt(ptab.desc(), ptr(x0p), ptr(tp), BP, LP, None, 1, 0, ptr(xtp), None, sp))
            check(lib.d3pm_lvb_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 0,
                                    1.0 / BP, ptr(perp), ptr(dpp), ptr(wsp), wsbp, sp))
        for _ in range(20):
            lvb_step()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(100):
            lvb_step()
        e1.record()
        torch.cuda.synchronize()
        lms = e0.elapsed_time(e1) / 100
        lvb = {"workload": "protein-style step: batched forward corruption (device noise) + D3PMLVBLoss forward and "
                           "gradient, B=256 L=512 K=22 n_tokens=24, float64 arithmetic, back-to-back launches",
               "ms_per_step": lms, "tokens_per_s": BP * LP / (lms * 1e-3)}

    if rank == 0:
        achieved = BYTES_PER_TOKEN * B * L / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic protein D3PM B={B} L={L} K={K} T={T} {args.kind} Q, per GPU: timestep draw + "
                                   "q_sample (Philox) + fused posterior/hybrid-loss fwd+bwd to per-example losses and dlogits (of the batch mean), "
                                   "denoiser excluded",
                       "global_batch": B * world, "seq_len": L, "parallelism": f"batch-sharded x{world}, no collective",
                       "launch": "eager" if args.no_graph else f"hip-graph replay, {spg} step(s) per graph",
                       "one_step_per_graph_ms_per_step": single_ms,
                       "calls": "training_losses(...).mean().backward()" if args.compat else "training_step(...)",
                       "other_call_shape_ms_per_step": other_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "kernel": "k_hybrid_loss via d3pm_hybrid_loss, back-to-back launches between two HIP events",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_token": BYTES_PER_TOKEN},
            "reverse_sample": None if args.no_secondary else {"value": done * world / rev_el, "unit": "seqs/s",
                               "what": "full T=500 (499 steps) reverse sampling, L=263, K=22, 10000 sequences sharded "
                                       f"over {world} GPU(s) in sub-batches of {sub}, the loop of each sub-batch replayed as one HIP "
                                       "graph, path only (frozen logits)",
                               "end_to_end_tiny_denoiser": done_e2e * world / e2e_el},
            "image_k256": dense,
            "protein_lvb": lvb,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
