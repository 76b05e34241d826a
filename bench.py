#!/usr/bin/env python3
"""Benchmark of the D3PM hot path on MI355X.

One step = one pass of the hot path over one synthetic batch of the protein
configuration (B=256, L=512, K=32, T=500, uniform transitions): timestep draw +
forward corruption `q_sample` (on-device Philox noise) + the fused posterior /
hybrid loss forward+backward kernel producing per-example losses and
d mean / d logits, handed to autograd.  The denoiser is excluded: its logits are
a pre-filled device tensor.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script starts N fresh child processes
itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before it
touches a GPU, relays rank 0's JSON line and exits with the first non-zero child
status; under `python -m torch.distributed.run` it is one of the ranks.  The
batch is sharded with no data-path collective (weak scaling: every rank runs the
full per-GPU batch); the process group (RCCL) carries the barriers and the
max-over-ranks time only.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
MIN_TIMED_S = 5e-3             # the timed region lasts at least this long whatever --steps says
N_SEQS, SEQ_LEN, K_PROT, T_PROT = 10000, 263, 22, 500
SUB_BATCH = 1250               # reverse sampling: sequences per graph replay (10 000 / 8 GPUs)
# HBM bytes per launch of a kernel from the PMC passes kept under profiles/ (2 x FETCH_SIZE + WRITE_SIZE, gfx950
# correction); the newest committed profile that names the kernel wins.  Not measured by this run: counters need a
# rocprofv3 pass of their own.
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r03_kernels_pmc_hbm.json", "r02_kernels_pmc_hbm.json",
                                                             "r01_bench_pmc_hbm.json")]
# `rocprofv3 --kernel-trace --stats` of `bench.py --roofline-only` (the rotating-buffer launch sequence alone)
TRACE_FILES = [os.path.join(ROOT, "profiles", f) for f in ("r03_roofline_kernel_stats.csv",)]


# the headline kernel: gradient, no next-batch corruption, one pass (the workgroup covers its batch element)
HEAD_KERNEL = "k_hybrid_loss_pair32<true, false, true>"


def measured_traffic(kernel: str, must_contain: str = ""):
    for path in TRAFFIC_FILES:
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, ctr in prof.items():
            if kernel in name and must_contain in name[name.index(kernel):] \
                    and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
                return {"bytes": (2 * ctr["FETCH_SIZE"]["mean_per_launch"] + ctr["WRITE_SIZE"]["mean_per_launch"]) * 1024,
                        "source": os.path.relpath(path, ROOT)}
    return None


def traced_duration_ms(kernel: str):
    """Average duration of `kernel` in the newest committed rocprofv3 kernel trace of `bench.py --roofline-only`."""
    import csv
    for path in TRACE_FILES:
        try:
            rows = list(csv.DictReader(open(path)))
        except OSError:
            continue
        for r in rows:
            if kernel in r.get("Name", ""):
                return {"ms": float(r["AverageNs"]) * 1e-6, "calls": int(r["Calls"]), "source": os.path.relpath(path, ROOT)}
    return None


# ------------------------------------------------------------------ N-rank launch
def spawn_plan(n: int, master_port: int, base_env=None):
    """Environment of each of the n ranks `bench.py --gpus n` starts: one process per GPU of this node."""
    base = dict(os.environ if base_env is None else base_env)
    plan = []
    for r in range(n):
        env = dict(base)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        plan.append(env)
    return plan


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Start n fresh children of this script (the parent has made no GPU call), relay rank 0's stdout, return the
    first non-zero exit status (0 if every rank succeeded).  A failing rank ends the others."""
    import threading
    plan = spawn_plan(n, _free_port())
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL)
             for r, env in enumerate(plan)]
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                 # exactly the children started above, by PID
                    procs[q].terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode(errors="replace"))
    sys.stdout.flush()
    return rc


# ------------------------------------------------------------------ CPU baselines
def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(step, budget_s: float, warm: int = 3, blocks: int = 10):
    """`warm` untimed calls, then the median over `blocks` timed blocks; a block repeats the call so that the
    blocks together take about `budget_s`."""
    for _ in range(warm):
        step()
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    per_block = max(1, int(budget_s / blocks / max(one, 1e-6)))
    times = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(per_block):
            step()
        times.append((time.perf_counter() - t0) / per_block)
    times.sort()
    return 0.5 * (times[blocks // 2 - 1] + times[blocks // 2]), per_block * blocks


def cpu_baselines(budget_s: float = 10.0, image_tables=None):
    """The CPU oracle (a restatement of the reference's eager PyTorch path: float64 tables cast per call, every
    [N,K] intermediate materialised, per-sequence Python loops in the protein style, autograd backward) timed on
    this host's cores: the headline step, one protein-style loss step, one reverse-sampling step."""
    import torch
    from oracle import image_style as im
    from oracle import protein_style as pr
    cores = torch.get_num_threads()
    host = f"{_cpu_model()}, torch {torch.__version__} CPU, {cores} threads; 3 warm-ups, median of 10 blocks"
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

    med, n = _median_time(step, budget_s)
    head = {"value": B * L / med, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of the same workload (B={B}, L={L}, K={K}, T={T}, uniform, hybrid loss fwd+bwd incl. "
                      f"q_sample and noise draw), median {med * 1e3:.1f} ms per step; {host}"}

    # protein-style loss: D3PMLVBLoss forward + backward on a 64-sequence slice of the B=256 L=512 n_tokens=24 batch
    BS, LP, LDP = 64, 512, 24
    qb, qt = pr.uniform_tables(K_PROT, T_PROT, "sohl-dickstein")
    tg = torch.randint(0, K_PROT, (BS, LP), generator=g)
    src = torch.randint(0, K_PROT, (BS, LP), generator=g)
    tsp = torch.randint(2, T_PROT, (BS,), generator=g)
    pred = torch.randn(BS, LP, LDP, generator=g, requires_grad=True)
    src_oh, tg_oh = pr.one_hot(src, K_PROT), pr.one_hot(tg, K_PROT)

    def lvb_step():
        pred.grad = None
        pr.driver_loss(src_oh, None, pred, tg, tg_oh, tsp, qt, qb, LAMBDA, K=K_PROT, tmax=T_PROT)[0].backward()

    med_l, n_l = _median_time(lvb_step, budget_s)
    lvb = {"value": BS * LP / med_l, "unit": "tokens/s", "cores": cores, "kind": "port",
           "sample": f"{n_l} steps of (D3PMLVBLoss + {LAMBDA} * CrossEntropyLoss) * n_tokens forward + backward (the "
                     f"training loop's loss) on a {BS}-sequence slice (L={LP}, K={K_PROT}, n_tokens={LDP}, "
                     f"2 <= t < {T_PROT}), median {med_l * 1e3:.1f} ms per step; {host}"}

    # reverse sampling: one step of generate_d3pm on a 256-sequence slice, extrapolated to the 499 steps
    BR = 256
    xr = torch.randint(0, K_PROT, (BR, SEQ_LEN), generator=g)
    lgr = torch.randn(BR, SEQ_LEN, K_PROT, generator=g)

    def rev_step():
        e = torch.empty(BR, SEQ_LEN, K_PROT, dtype=torch.float64).exponential_(1)
        pr.reverse_step(lgr, xr, 250, qt, qb, e, K_PROT)

    med_r, n_r = _median_time(rev_step, budget_s / 2)
    rev = {"value": BR / (med_r * (T_PROT - 1)), "unit": "seqs/s", "cores": cores, "kind": "port",
           "sample": f"{n_r} reverse steps (posterior marginal + multinomial draw, frozen logits) on a {BR}-sequence "
                     f"slice (L={SEQ_LEN}, K={K_PROT}), median {med_r * 1e3:.1f} ms per step, extrapolated to the "
                     f"{T_PROT - 1} steps of a full run; {host}"}
    # image configuration (K=256, T=1000, Gaussian): the oracle's training step on an 8-image slice of the 128-image
    # batch.  The [T,K,K] float64 tables are the ones the GPU built (the reference's own build is 21.8 s of host
    # time and is not part of a step); the step casts them whole to float32 per gather, as the reference does.
    img = None
    if image_tables is not None:
        q_mats, qT, betas_i = image_tables
        pi = im.Process(betas=betas_i, kind="gaussian", K=256, bands=None, q_mats=q_mats, qT_onestep=qT,
                        hybrid_coeff=LAMBDA, loss_type="hybrid")
        BI = 8
        x0i = torch.randint(0, 256, (BI, 32, 32, 1), generator=g)
        lgi = torch.randn(BI, 32, 32, 1, 256, generator=g, requires_grad=True)

        def img_step():
            noise = torch.rand(BI, 32, 32, 1, 256)
            t = torch.randint(0, 1000, (BI,), dtype=torch.int32)
            loss, _ = im.training_step(pi, lambda x, tt: lgi, x0i, t, noise)
            lgi.grad = None
            loss.mean().backward()

        med_i, n_i = _median_time(img_step, budget_s, warm=1, blocks=4)
        img = {"value": BI * 1024 / med_i, "unit": "tokens/s", "cores": cores, "kind": "port",
               "sample": f"{n_i} training steps (q_sample + hybrid loss fwd+bwd) on an {BI}-image slice of the batch "
                         f"(32x32x1, K=256, T=1000, Gaussian Q), median {med_i * 1e3:.0f} ms per step; {host}"}
    return head, lvb, rev, img


# ------------------------------------------------------------------ one rank
def worker(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    t_start = time.perf_counter()

    def progress(what):
        """One stderr line per section when D3PM_BENCH_PROGRESS is set (a run under a profiler shows where it is)."""
        if os.environ.get("D3PM_BENCH_PROGRESS") and rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {what}", file=sys.stderr, flush=True)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch `python bench.py --gpus N` "
                         "(it starts the ranks itself) or torchrun with --nproc-per-node equal to --gpus")
    # the process group (RCCL) is only needed for N > 1; D3PM_BENCH_FORCE_PG=1 creates it for a single rank too, to
    # rehearse the barrier / all-reduce calls of the multi-GPU run on a one-GPU box
    use_pg = world > 1 or bool(os.environ.get("D3PM_BENCH_FORCE_PG"))
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node shows "
                         f"{torch.cuda.device_count()} device(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (the process group is created after every HIP graph below has been captured: no communicator thread is
    # alive while a stream is capturing)

    import d3pm_amd
    from d3pm_amd.sharding import shard_range, equal_sub_batches
    from d3pm_amd.tables import protein_tables
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS

    torch.manual_seed(1234 + rank)
    betas = d3pm_amd.generate_betas(0.0, 0.0, T, "jsd")
    x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
    logits = torch.randn(B, L, 1, 1, K, device=dev, requires_grad=True)
    model_fn = lambda x, tt: logits
    sp = stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def process(kind):
        d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=5, transition_bands=None,
                                       model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                       hybrid_coeff=LAMBDA, device=dev)
        assert d.tables().structured
        return d

    d = process(args.kind)

    def event_ms(fn, n, warm=10):
        """Average duration of fn's launches back to back on the launch stream, between two HIP events."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        until = time.perf_counter() + 0.1            # sustained clocks before anything is timed
        while time.perf_counter() < until:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def event_ms_replayed(fn_on, n, per_graph=50):
        """The same figure with the host out of the way: `per_graph` launches of fn_on(stream pointer) captured back to
        back into a HIP graph, replays timed between two HIP events on the replay stream (a Python call per launch
        takes about as long as these kernels run, so eager back-to-back timing sits at the edge of being host-bound)."""
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                fn_on(stream_ptr())
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            spc = stream_ptr()
            for _ in range(per_graph):
                fn_on(spc)
        reps = max(2, -(-n // per_graph))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * per_graph)

    # ---- duration of the dominant kernel: the same launch (same buffers) back to back on the stream, HIP events
    # at both ends; rocprofv3 --kernel-trace of this command reports the same kernel's average duration (profiles/).
    def fused_kernel_ms(dd, nsets=1, with_next=False, replayed=False, copy=False):
        """nsets > 1 rotates through that many disjoint sets of inputs and outputs (> 256 MiB in all), so that no
        launch finds its rows in the Infinity Cache: that figure is HBM traffic proper."""
        tb = dd.tables()
        t_s = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
        sets = []
        for i in range(nsets):
            x0_i = x0 if i == 0 else torch.randint(0, K, (B, L, 1, 1), device=dev)
            lg_i = logits.detach() if i == 0 else torch.randn(B, L, 1, 1, K, device=dev)
            sets.append((x0_i, dd.q_sample(x0_i, t_s), lg_i, torch.empty_like(lg_i)))
        loss_s = torch.empty(B, device=dev)
        wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        state = {"i": 0}

        def fused_only(spx=None):
            x0_i, xt_i, lg_i, dl_i = sets[state["i"] % nsets]
            state["i"] += 1
            check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg_i), K, ptr(x0_i), 8, ptr(xt_i), 8, ptr(t_s), 4, B, L,
                                       LOSS["hybrid"], LAMBDA, 1.0 / B, None, None, None, ptr(loss_s), ptr(dl_i), None,
                                       None, 0, ptr(ws), wsb, sp if spx is None else spx))
        if copy:
            # a plain device copy over the same buffers (torch's elementwise copy kernel, dlogits <- logits: 8K of the
            # launch's 8K + 16 bytes per token, no arithmetic): what the memory system gives a streaming kernel of
            # this size and access pattern
            def copy_only():
                _, _, lg_i, dl_i = sets[state["i"] % nsets]
                state["i"] += 1
                dl_i.copy_(lg_i)
            blocks = sorted(event_ms(copy_only, max(args.steps, 200), warm=2 * nsets + 10) for _ in range(5))
            return blocks[2]
        if replayed:
            return event_ms_replayed(fused_only, max(args.steps, 200), per_graph=10 * nsets if nsets > 1 else 50)
        if not with_next:
            # median of five blocks of back-to-back launches (a block that another tenant of the box disturbs does
            # not decide the figure)
            blocks = sorted(event_ms(fused_only, max(args.steps, 200), warm=2 * nsets + 10) for _ in range(5))
            return blocks[2]
        t_n = torch.empty(B, dtype=torch.int32, device=dev)
        xt_n = torch.empty(B, L, 1, 1, dtype=torch.int64, device=dev)

        def fused_with_next():
            x0_i, xt_i, lg_i, dl_i = sets[0]
            check(lib.d3pm_hybrid_loss_then_q_sample(tb.desc(), ptr(lg_i), K, ptr(x0_i), 8, ptr(xt_i), 8, ptr(t_s), 4, B, L,
                                                     LOSS["hybrid"], LAMBDA, 1.0 / B, None, None, None, ptr(loss_s),
                                                     ptr(dl_i), ptr(ws), wsb, ptr(x0_i), 8, 7, 0, None, ptr(t_n),
                                                     ptr(xt_n), sp))
        return event_ms(fused_with_next, max(args.steps, 200), warm=12)


    if args.roofline_only:
        # exactly the launch sequence roofline.frac is quoted on, and nothing else of this kernel: the program of the
        # rocprofv3 kernel trace kept under profiles/ (its average duration of k_hybrid_loss_pair32<true, false, true> is
        # this figure as the tracer sees it)
        ms = fused_kernel_ms(d, nsets=10)
        ach = BYTES_PER_TOKEN * B * L / (ms * 1e-3) / 1e9
        print(json.dumps({"roofline_only": True, "kernel": HEAD_KERNEL, "kernel_ms": ms,
                          "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                          "launches": "10 rotating buffer sets, back to back between two HIP events, median of 5 blocks"}),
              flush=True)
        return

    # Default: DiscreteDiffusion.training_step -- timestep draw + q_sample (one launch, device Philox), fused loss
    # per example and d mean / d logits (one launch), gradient handed to autograd -- replayed as a HIP graph.
    # --compat: the reference-shaped calls, training_losses(...).mean().backward(), which add the caller's
    # reduction and its backward (four more short launches on [B] tensors).
    compat_steps = 1                       # steps per graph of the reference-shaped call sequence

    def make_step(dd, compat, graph):
        if compat and not graph:
            def step():
                t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
                x_t = dd.q_sample(x0, t)                              # Philox noise on device
                losses = dd._fused(logits, x0, x_t, t, "hybrid")     # fused fwd + bwd kernel (C ABI d3pm_hybrid_loss)
                logits.grad = None
                losses.mean().backward()                              # hands dlogits to autograd (early-exit rescale)
            return step
        if compat:
            return d3pm_amd.graphs.GraphedTrainingLosses(dd, model_fn, x0, seed=1234 + rank, params=[logits],
                                                         steps=compat_steps).replay
        if not graph:
            def step():
                logits.grad = None
                dd.training_step(model_fn, x_start=x0)
            return step
        return d3pm_amd.graphs.GraphedTrainingStep(dd, model_fn, x0, seed=1234 + rank, params=[logits]).replay

    # n consecutive steps per graph replay (default step shape only): K timed steps = K // n replays of the n-step
    # graph + K % n replays of the one-step graph
    spg = args.steps_per_graph if not (args.compat or args.no_graph) else 1

    pipelined = spg > 1 and not args.no_pipeline

    def make_runner(dd, seed):
        one = make_step(dd, args.compat, not args.no_graph)
        multi = (d3pm_amd.graphs.GraphedTrainingStep(dd, model_fn, x0, seed=seed, params=[logits], steps=spg,
                                                     pipelined=pipelined).replay
                 if spg > 1 else None)

        # the steps left over when --steps is no multiple of n: one graph of that many steps (captured here, before
        # the process group exists), so that they are not timed at one host-bound graph launch per step
        rem = args.steps % spg if spg > 1 else 0
        rest = (d3pm_amd.graphs.GraphedTrainingStep(dd, model_fn, x0, seed=seed + 1, params=[logits], steps=rem,
                                                    pipelined=pipelined).replay if rem > 1 else None)

        def run_steps(k):
            if multi is not None:
                for _ in range(k // spg):
                    multi()
                k %= spg
            if rest is not None and k == rem:
                rest()
                k = 0
            for _ in range(k):
                one()
        return run_steps, one

    run_steps, step = make_runner(d, 4321 + rank)
    # for the record: the same n-step graph with two launches per step (captured here: before the process group)
    plain_multi = (d3pm_amd.graphs.GraphedTrainingStep(d, model_fn, x0, seed=99 + rank, params=[logits], steps=spg).replay
                   if pipelined and not args.no_secondary else None)
    # the call shape that is not timed, for the record, at the same number of steps per graph as the timed one
    compat_steps = spg if (not args.compat and not args.no_graph) else 1
    other = make_step(d, not args.compat, not args.no_graph) if not args.no_secondary else None
    other_steps = compat_steps if not args.compat else 1
    compat_steps = 1
    d_abs = run_abs = None
    if not args.no_secondary and rank == 0:
        d_abs = process("absorbing" if args.kind == "uniform" else "uniform")
        run_abs, _ = make_runner(d_abs, 987 + rank)

    # secondary: reverse sampling of 10k BLAT-length sequences, sharded; every rank takes its contiguous share and
    # cuts it into equal sub-batches of at most SUB_BATCH sequences (one HIP graph replay of the 499-step loop each)
    samplers, sub_sizes, ptab = None, [], None
    if not args.no_secondary:
        lo, hi = shard_range(N_SEQS, rank, world)
        sub_sizes = equal_sub_batches(hi - lo, SUB_BATCH)
        tok = d3pm_amd.Tokenizer()
        Qb, Qt = tok.q_uniform_schedule(num_steps=T_PROT, device=dev)
        ptab = protein_tables(Qt, Qb)
        frozen = torch.randn(max(sub_sizes), SEQ_LEN, K_PROT, device=dev)
        emb = torch.randn(K_PROT, 64, device=dev) * 0.5
        tvec = torch.randn(T_PROT + 1, 64, device=dev) * 0.1
        head = torch.randn(64, 24, device=dev) * 0.3
        tiny = lambda x, tt: (emb[x] + tvec[tt][:, None, :]) @ head
        frozen_fn = lambda x, tt: frozen[:x.shape[0]]
        # [path only, with a tiny denoiser] x [distinct sub-batch sizes]
        samplers = [{nb: d3pm_amd.graphs.GraphedReverseSampler(ptab, fn, nb, SEQ_LEN, T_PROT, dev,
                                                                seed=77 + rank + 1000 * k,
                                                                capture=not args.no_graph)
                     for k, nb in enumerate(sorted(set(sub_sizes))) if nb > 0} for fn in (frozen_fn, tiny)]
    torch.cuda.synchronize()
    ranks_seen = 1
    if use_pg:
        # RCCL prints a version banner on file descriptor 1 when its first communicator comes up: send it to
        # stderr, so that stdout carries the JSON line only
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            os.dup2(keep, 1)
            os.close(keep)
        ranks_seen = dist.get_world_size()
        assert ranks_seen == world, (ranks_seen, world)

    def timed_steps(run, k, warm):
        """W warm-up steps, then K x reps steps between barriers; reps makes the region last >= MIN_TIMED_S."""
        # bring the GPU to its sustained clocks first (a step is ~15 us: a short run would otherwise be timed on a
        # part that has just left idle)
        ramp_until = time.perf_counter() + 0.2
        while time.perf_counter() < ramp_until:
            run(48)
            torch.cuda.synchronize()
        run(warm)
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        run(k)
        torch.cuda.synchronize()
        probe = max(time.perf_counter() - p0, 1e-6)
        reps = max(1, int(-(-MIN_TIMED_S // probe)))
        if use_pg:                       # the same repeat count on every rank
            r_ = torch.tensor([reps], dtype=torch.int64, device=dev)
            dist.all_reduce(r_, op=dist.ReduceOp.MAX)
            reps = int(r_.item())
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run(k)
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_pg:
            el_ = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(el_, op=dist.ReduceOp.MAX)
            el = float(el_.item())
        return el, reps

    progress("timed steps")
    elapsed, reps = timed_steps(run_steps, args.steps, args.warmup)
    n_timed = args.steps * reps
    tokens = B * L * n_timed * world

    progress("single-step and other call shape")
    single_ms = None
    if spg > 1:                     # for the record: one step per graph replay (host-launch-bound)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - s0) * 5.0
    plain_ms = None
    if plain_multi is not None:     # two launches per step, same number of steps per graph
        for _ in range(10):
            plain_multi()
        torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(50):
            plain_multi()
        torch.cuda.synchronize()
        plain_ms = (time.perf_counter() - s0) / (50 * spg) * 1e3
    other_ms = None
    if other is not None:           # the other call shape, for the record (same graph/eager mode, 100 steps)
        for _ in range(10):
            other()
        torch.cuda.synchronize()
        o0 = time.perf_counter()
        for _ in range(100):
            other()
        torch.cuda.synchronize()
        other_ms = (time.perf_counter() - o0) * 10.0 / other_steps

    progress("fused kernel back to back")
    kern_ms = fused_kernel_ms(d)
    kern_rot_ms = fused_kernel_ms(d, nsets=10)
    copy_rot_ms = fused_kernel_ms(d, nsets=10, copy=True)
    copy_ms = fused_kernel_ms(d, copy=True)
    kern_next_ms = fused_kernel_ms(d, with_next=True) if pipelined else None
    kern_replayed_ms = fused_kernel_ms(d, replayed=True) if not args.no_graph and not use_pg else None

    progress("absorbing")
    absorbing = None
    if run_abs is not None:
        for _ in range(3):
            run_abs(48)
        torch.cuda.synchronize()
        a0 = time.perf_counter()
        run_abs(400)
        torch.cuda.synchronize()
        a_ms = (time.perf_counter() - a0) / 400 * 1e3
        ak = fused_kernel_ms(d_abs)
        a_ach = BYTES_PER_TOKEN * B * L / (ak * 1e-3) / 1e9
        absorbing = {"workload": f"the same step with {d_abs.transition_mat_type} Q (this rank only)",
                     "ms_per_step": a_ms, "tokens_per_s": B * L / (a_ms * 1e-3),
                     "roofline": {"bound": "hbm", "achieved": a_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": a_ach / HBM_PEAK_GBS, "kernel_ms": ak}}

    # ---- secondary: reverse sampling
    progress("reverse sampling")
    reverse = None
    if samplers is not None:
        def run_sampler(smps, passes):
            """This rank's share of the 10 000 sequences through the 499-step loop, one HIP graph replay per
            sub-batch.  The whole job is timed `passes` times back to back, every pass between a barrier and a
            device synchronisation and taken as the slowest rank's time; the median pass is reported (at 8 ranks a
            pass is ~3 ms per rank: one pass alone would be at the mercy of the barrier's skew)."""
            for smp in smps.values():
                smp.sample()
            torch.cuda.synchronize()
            times, n = [], 0
            for _ in range(passes):
                if use_pg:
                    dist.barrier()
                torch.cuda.synchronize()
                r0 = time.perf_counter()
                n = 0
                for nb in sub_sizes:
                    if nb > 0:
                        smps[nb].sample()
                        n += nb
                torch.cuda.synchronize()
                times.append(time.perf_counter() - r0)
            if use_pg:
                mx = torch.tensor(times, dtype=torch.float64, device=dev)
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                times = mx.tolist()
                nn_ = torch.tensor([float(n)], dtype=torch.float64, device=dev)
                dist.all_reduce(nn_, op=dist.ReduceOp.SUM)
                n = int(round(float(nn_.item())))
            times.sort()
            return n, times[len(times) // 2], times

        REV_PASSES, E2E_PASSES = 5, 3
        done, rev_el, rev_all = run_sampler(samplers[0], REV_PASSES)
        done_e2e, e2e_el, _ = run_sampler(samplers[1], E2E_PASSES)
        # the reverse-step kernel alone, at the largest sub-batch of this rank
        nb = max(sub_sizes)
        xr = torch.randint(0, K_PROT, (nb, SEQ_LEN), device=dev)
        tr = torch.full((nb,), 250, dtype=torch.long, device=dev)
        outr = torch.empty_like(xr)
        frz = torch.randn(nb, SEQ_LEN, K_PROT, device=dev)

        def rev_only():
            check(lib.d3pm_lvb_reverse_step(ptab.desc(), ptr(frz), K_PROT, ptr(xr), ptr(tr), nb, SEQ_LEN, None, 7, 0,
                                            None, 1.0, 0, ptr(outr), sp))
        rk_ms = event_ms(rev_only, 200)
        rbytes = 4 * K_PROT + 16
        r_ach = rbytes * nb * SEQ_LEN / (rk_ms * 1e-3) / 1e9
        reverse = {"value": done / rev_el, "unit": "seqs/s",
                   "what": f"full T={T_PROT} ({T_PROT - 1} steps) reverse sampling, L={SEQ_LEN}, K={K_PROT}, {N_SEQS} "
                           f"sequences sharded over {world} GPU(s), every rank its contiguous share in sub-batches of "
                           f"{sorted(set(sub_sizes), reverse=True)} sequences, the loop of each sub-batch "
                           + ("launched step by step" if args.no_graph else "replayed as one HIP graph")
                           + ", path only (frozen logits)",
                   "sequences": done, "seconds": rev_el,
                   "timing": f"median of {REV_PASSES} passes over the whole job (slowest rank per pass); "
                             f"end to end: median of {E2E_PASSES}",
                   "seconds_all_passes": rev_all,
                   "end_to_end_tiny_denoiser": done_e2e / e2e_el,
                   "roofline": {"bound": "hbm", "achieved": r_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": r_ach / HBM_PEAK_GBS, "kernel": "k_lvb_reverse_flat<22, 22> via "
                                "d3pm_lvb_reverse_step, back-to-back launches between two HIP events",
                                "kernel_ms": rk_ms, "algorithmic_bytes_per_token": rbytes,
                                "tokens_per_launch": nb * SEQ_LEN,
                                "traffic": measured_traffic("k_lvb_reverse_flat<22, 22>")}}

    # ---- secondary: the image configuration (B=128, 32x32x1, K=256, T=1000, Gaussian Q): dense contraction on
    # the matrix cores, fused loss fwd+bwd; priced against the dense float32 MFMA peak
    progress("image K=256")
    dense = None
    image_tables_cpu = None
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
        dms = event_ms(dense_only, 100, warm=20)
        tfl = 4.0 * KI * KI * BI * LI / (dms * 1e-3) / 1e12
        # the same launch on the float32 matrix instructions (dense_matmul='f32'): the exact-float32 contraction the
        # default emulates with two float16 pieces per operand and three products per term
        from d3pm_amd.tables import ImageTables
        tbf = ImageTables(di.q_mats, di.transpose_q_onestep_mats, eps=di.eps, transposed_onestep=True, dense_matmul="f32")

        def dense_f32():
            check(lib.d3pm_hybrid_loss(tbf.desc(), ptr(lgi), KI, ptr(x0i), 8, ptr(xti), 8, ptr(ti), 4, BI, LI,
                                       LOSS["hybrid"], LAMBDA, 1.0 / BI, None, None, None, ptr(lossi), ptr(dli), None, None, 0, ptr(wsi),
                                       wsbi, sp))
        fms = event_ms(dense_f32, 60, warm=12)
        tflf = 4.0 * KI * KI * BI * LI / (fms * 1e-3) / 1e12
        del tbf
        dense = {"workload": "image D3PM B=128 32x32x1 K=256 T=1000 Gaussian Q, fused loss fwd+bwd to dlogits",
                 "kernel_ms": dms, "tokens_per_s": BI * LI / (dms * 1e-3),
                 "roofline": {"bound": "mfma", "achieved": tfl, "peak": 157.3, "unit": "TFLOP/s", "frac": tfl / 157.3,
                              "flops_per_token": 4 * KI * KI,
                              "dtype": "f32 emulated on the float16 matrix pipe: operands scaled and split into two float16 pieces, three "
                                       "v_mfma_f32_16x16x32_f16 products per term (relative error < 2^-22 per product); "
                                       "achieved / peak are algorithmic float32 flops against the float32 MFMA peak",
                              "executed_f16_tflops": 3 * tfl, "executed_peak_f16": 2500.0,
                              "executed_frac_of_f16_peak": 3 * tfl / 2500.0},
                 "f32_matrix_instructions": {"what": "dense_matmul='f32': v_mfma_f32_32x32x2_f32, an in-order float32 fma chain",
                                             "kernel_ms": fms, "achieved": tflf, "peak": 157.3, "frac": tflf / 157.3}}
        # the sampling loop's reverse step on the same tables (device noise, no probabilities out), and the same two
        # launches on 7- and 6-bit data (K = 128 / 64, Gaussian Q): the register-tile kernels at their other sizes
        smpi = torch.empty(BI, 32, 32, 1, dtype=torch.int64, device=dev)

        def dense_sample():
            check(lib.d3pm_p_sample(tbi.desc(), ptr(lgi), KI, ptr(xti), 8, ptr(ti), 4, BI, LI, None,
                                    d3pm_amd._lib.NOISE_PHILOX, 1, 0, None, ptr(smpi), None, sp))
        dense["p_sample"] = {"what": "d3pm_p_sample, device noise, no probabilities out, same tokens",
                             "kernel_ms": event_ms(dense_sample, 100, warm=20)}
        narrower = {}
        for bits in (7, 6):
            Kn = 2 ** bits
            dn = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(1e-4, 0.02, TI, "linear"),
                                            transition_mat_type="gaussian", num_bits=bits, transition_bands=None,
                                            model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                            hybrid_coeff=LAMBDA, device=dev)
            tbn = dn.tables()
            x0n = torch.randint(0, Kn, (BI, 32, 32, 1), device=dev)
            xtn = dn.q_sample(x0n, ti)
            lgn = torch.randn(BI, 32, 32, 1, Kn, device=dev)
            dln = torch.empty_like(lgn)

            def loss_n():
                check(lib.d3pm_hybrid_loss(tbn.desc(), ptr(lgn), Kn, ptr(x0n), 8, ptr(xtn), 8, ptr(ti), 4, BI, LI,
                                           LOSS["hybrid"], LAMBDA, 1.0 / BI, None, None, None, ptr(lossi), ptr(dln), None,
                                           None, 0, ptr(wsi), wsbi, sp))

            def sample_n():
                check(lib.d3pm_p_sample(tbn.desc(), ptr(lgn), Kn, ptr(xtn), 8, ptr(ti), 4, BI, LI, None,
                                        d3pm_amd._lib.NOISE_PHILOX, 1, 0, None, ptr(smpi), None, sp))
            lms = event_ms(loss_n, 100, warm=20)
            narrower[f"K{Kn}"] = {"loss_fwd_bwd_ms": lms, "p_sample_ms": event_ms(sample_n, 100, warm=20),
                                  "loss_hbm_frac": (8 * Kn + 16) * BI * LI / (lms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del dn, tbn, lgn, dln
        dense["narrower_alphabets"] = narrower
        if not args.no_cpu_baseline and world == 1:
            image_tables_cpu = (di.q_mats.cpu(), di.transpose_q_onestep_mats.cpu(), di.betas.cpu())
        # three channels (the CIFAR shape of the reference's notebook): B=128 32x32x3, 3072 positions per image
        del lgi, dli
        L3 = 3072
        x03 = torch.randint(0, KI, (BI, 32, 32, 3), device=dev)
        xt3 = di.q_sample(x03, ti)
        lg3 = torch.randn(BI, 32, 32, 3, KI, device=dev)
        dl3 = torch.empty_like(lg3)
        wsb3 = lib.d3pm_hybrid_loss_workspace(BI, L3, KI)
        ws3 = torch.empty(max(wsb3 // 4, 1), device=dev)

        def dense_c3():
            check(lib.d3pm_hybrid_loss(tbi.desc(), ptr(lg3), KI, ptr(x03), 8, ptr(xt3), 8, ptr(ti), 4, BI, L3,
                                       LOSS["hybrid"], LAMBDA, 1.0 / BI, None, None, None, ptr(lossi), ptr(dl3), None, None, 0,
                                       ptr(ws3), wsb3, sp))
        d3ms = event_ms(dense_c3, 40, warm=8)
        tfl3 = 4.0 * KI * KI * BI * L3 / (d3ms * 1e-3) / 1e12
        dense["three_channels"] = {"workload": "the same with B=128 32x32x3 (393 216 positions per launch)",
                                   "kernel_ms": d3ms, "tokens_per_s": BI * L3 / (d3ms * 1e-3),
                                   "roofline": {"bound": "mfma", "achieved": tfl3, "peak": 157.3, "unit": "TFLOP/s",
                                                "frac": tfl3 / 157.3}}
        del di, tbi, lg3, dl3

    # ---- secondary: the protein-style training loss (D3PMLVBLoss forward + gradient) and the batched forward
    # corruption of the collaters, B=256 L=512 K=22 n_tokens=24, through the C ABI
    progress("protein LVB")
    lvb = None
    if rank == 0 and not args.no_secondary:
        BP, LP, KP, LDP = 256, 512, 22, 24
        x0p = torch.randint(0, KP, (BP, LP), device=dev)
        tp = torch.randint(2, 500, (BP,), device=dev)
        xtp = torch.empty_like(x0p)
        predp = torch.randn(BP, LP, LDP, device=dev)
        dpp = torch.empty_like(predp)
        perp = torch.empty(BP, dtype=torch.float64, device=dev)
        cep = torch.empty(BP, dtype=torch.float64, device=dev)
        wsbp = lib.d3pm_lvb_loss_workspace(BP, LP, KP)
        wsp = torch.empty(max(wsbp // 8, 1), dtype=torch.float64, device=dev)

        def lvb_corrupt():
            check(lib.d3pm_corrupt(ptab.desc(), ptr(x0p), ptr(tp), BP, LP, None, 1, 0, ptr(xtp), None, sp))

        def lvb_loss():
            # the training loop's loss (notebook cell-12): (lvb + lambda * ce) * n_tokens, n_tokens = B * L
            check(lib.d3pm_lvb_hybrid_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 0,
                                           float(LP), LAMBDA * LP, ptr(perp), ptr(cep), ptr(dpp), ptr(wsp), wsbp, sp))

        def lvb_only():
            check(lib.d3pm_lvb_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 0,
                                    1.0 / BP, ptr(perp), ptr(dpp), ptr(wsp), wsbp, sp))

        def lvb_step():
            lvb_corrupt()
            lvb_loss()
        lms = event_ms(lvb_step, 200, warm=20)
        lk_ms = event_ms(lvb_loss, 200)
        lo_ms = event_ms(lvb_only, 200)
        lbytes = 8 * LDP + 16
        l_ach = lbytes * BP * LP / (lk_ms * 1e-3) / 1e9
        lvb = {"workload": "protein-style step: batched forward corruption (device noise) + the training loop's loss, "
                           "LVB + CE: (D3PMLVBLoss + 0.001 * CrossEntropyLoss) * n_tokens, forward and gradient in one "
                           "launch, B=256 L=512 K=22 n_tokens=24, back-to-back launches",
               "ms_per_step": lms, "tokens_per_s": BP * LP / (lms * 1e-3),
               "roofline": {"bound": "hbm", "achieved": l_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": l_ach / HBM_PEAK_GBS, "kernel": "k_lvb_loss_tpl<22, 24, false, true> via "
                            "d3pm_lvb_hybrid_loss, back-to-back launches between two HIP events", "kernel_ms": lk_ms,
                            "kernel_ms_lvb_alone": lo_ms, "algorithmic_bytes_per_token": lbytes,
                            "traffic": measured_traffic("k_lvb_loss_tpl<22, 24, false, true>")}}

    # ---- secondary: the two loss kernels at eight times the batch (B = 2048: the global batch of the DDP
    # configuration on one GPU; 285 / 218 MB per launch, more than the 256 MiB Infinity Cache holds), where launch
    # overhead and the arrival of the first rows are a small part of the launch
    progress("large batch")
    large = None
    if rank == 0 and not args.no_secondary:
        BB = 2048
        x0b = torch.randint(0, K, (BB, L, 1, 1), device=dev)
        tbg = torch.randint(0, T, (BB,), dtype=torch.int32, device=dev)
        xtb = d.q_sample(x0b, tbg)
        lgb = torch.randn(BB, L, 1, 1, K, device=dev)
        dlb = torch.empty_like(lgb)
        lsb = torch.empty(BB, device=dev)
        wsbb = lib.d3pm_hybrid_loss_workspace(BB, L, K)
        wsb_ = torch.empty(max(wsbb // 4, 1), device=dev)
        tbd = d.tables()

        def big_hybrid():
            check(lib.d3pm_hybrid_loss(tbd.desc(), ptr(lgb), K, ptr(x0b), 8, ptr(xtb), 8, ptr(tbg), 4, BB, L,
                                       LOSS["hybrid"], LAMBDA, 1.0 / BB, None, None, None, ptr(lsb), ptr(dlb), None, None, 0,
                                       ptr(wsb_), wsbb, sp))
        bh_ms = event_ms(big_hybrid, 50, warm=5)
        del lgb, dlb
        BPb, LPb, KPb, LDPb = BB, 512, 22, 24
        x0q = torch.randint(0, KPb, (BPb, LPb), device=dev)
        tq = torch.randint(2, 500, (BPb,), device=dev)
        xtq = torch.empty_like(x0q)
        check(lib.d3pm_corrupt(ptab.desc(), ptr(x0q), ptr(tq), BPb, LPb, None, 1, 0, ptr(xtq), None, sp))
        pq = torch.randn(BPb, LPb, LDPb, device=dev)
        dq = torch.empty_like(pq)
        perq = torch.empty(BPb, dtype=torch.float64, device=dev)
        wsbq = lib.d3pm_lvb_loss_workspace(BPb, LPb, KPb)
        wsq = torch.empty(max(wsbq // 8, 1), dtype=torch.float64, device=dev)

        ceq = torch.empty(BPb, dtype=torch.float64, device=dev)

        def big_lvb():
            check(lib.d3pm_lvb_hybrid_loss(ptab.desc(), ptr(pq), LDPb, ptr(x0q), ptr(xtq), ptr(tq), None, BPb, LPb, 500,
                                           0, float(LPb), LAMBDA * LPb, ptr(perq), ptr(ceq), ptr(dq), ptr(wsq), wsbq, sp))
        bl_ms = event_ms(big_lvb, 50, warm=5)
        del pq, dq

        def roof(bytes_per_token, ntok, ms, kernel):
            ach = bytes_per_token * ntok / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "kernel": kernel, "kernel_ms": ms, "algorithmic_bytes_per_token": bytes_per_token,
                    "tokens_per_launch": ntok}
        large = {"workload": f"the two loss kernels at B={BB}, L=512 (1 048 576 tokens per launch), back-to-back launches "
                             "between two HIP events",
                 "hybrid_loss": roof(BYTES_PER_TOKEN, BB * L, bh_ms, "k_hybrid_loss_tpl<32, true, 512> in 256-thread workgroups via d3pm_hybrid_loss"),
                 "lvb_loss": roof(8 * LDPb + 16, BPb * LPb, bl_ms, "k_lvb_loss_tpl<22, 24, false, true> (LVB + CE) via d3pm_lvb_hybrid_loss")}

    progress("sections done")
    if rank == 0:
        # the roofline figure is the launch over ten rotating buffer sets (357 MB: every launch reads its rows from
        # HBM proper); the same launch on one 35.7 MB working set, which the 256 MiB Infinity Cache serves, is kept
        # beside it as *_cache_resident
        achieved = BYTES_PER_TOKEN * B * L / (kern_rot_ms * 1e-3) / 1e9
        traffic = measured_traffic(HEAD_KERNEL)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": (traffic["source"] + " (a committed rocprofv3 --pmc pass of this command, not "
                                   "measured by this run)") if traffic else None,
                "kernel": HEAD_KERNEL + " via d3pm_hybrid_loss, back-to-back launches over 10 "
                          "rotating sets of inputs and outputs between two HIP events",
                "kernel_ms": kern_rot_ms, "algorithmic_bytes_per_token": BYTES_PER_TOKEN,
                "tokens_per_launch": B * L,
                "kernel_ms_cache_resident": kern_ms,
                "frac_cache_resident": BYTES_PER_TOKEN * B * L / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "frac / kernel_ms: 10 disjoint buffer sets (357 MB) in rotation, i.e. rows served from HBM; "
                        "*_cache_resident: one 35.7 MB working set relaunched back to back, which stays in the "
                        "256 MiB Infinity Cache; `bench.py --roofline-only` launches exactly the rotating sequence "
                        "(the program of the committed rocprofv3 kernel trace)"}
        roof["plain_copy_same_buffers"] = {
            "what": "torch copy_ of the logits tensor into the gradient tensor over the same buffer sets (8K of the kernel's "
                    "8K + 16 bytes per token, no arithmetic), timed the same way",
            "ms_rotating": copy_rot_ms, "ms_cache_resident": copy_ms,
            "GBps_rotating": 8 * K * B * L / (copy_rot_ms * 1e-3) / 1e9,
            "GBps_cache_resident": 8 * K * B * L / (copy_ms * 1e-3) / 1e9,
            "kernel_over_copy_rotating": kern_rot_ms / copy_rot_ms}
        traced = traced_duration_ms(HEAD_KERNEL)
        if traced:
            # the tracer serialises the dispatches and brackets each with its own packets: a kernel of this size
            # reads 0.4-0.5 us longer under it than its back-to-back rate, which is what the events (and the wall
            # clock of the replayed graph) see; both are given so that either division can be checked
            roof["kernel_ms_rocprof_trace"] = traced["ms"]
            roof["frac_rocprof_trace"] = BYTES_PER_TOKEN * B * L / (traced["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            roof["rocprof_trace_source"] = (f"{traced['source']}: {traced['calls']} launches of `bench.py "
                                            "--roofline-only` under rocprofv3 --kernel-trace (a committed trace, not "
                                            "measured by this run)")
        if kern_replayed_ms:
            roof["kernel_ms_graph_replayed"] = kern_replayed_ms
            roof["frac_graph_replayed"] = BYTES_PER_TOKEN * B * L / (kern_replayed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if kern_next_ms:
            # the launch the pipelined step actually makes: the same kernel carrying the next batch's timestep draw
            # and corruption (k_hybrid_loss_pair32<true, true, true>): 16 more bytes per token
            roof["kernel_ms_with_next_q_sample"] = kern_next_ms
            roof["frac_with_next_q_sample"] = (BYTES_PER_TOKEN + 16) * B * L / (kern_next_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": METRIC, "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / n_timed * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic protein D3PM B={B} L={L} K={K} T={T} {args.kind} Q, per GPU: timestep draw + "
                                   "q_sample (Philox) + fused posterior/hybrid-loss fwd+bwd to per-example losses and dlogits (of the batch mean), "
                                   "denoiser excluded",
                       "global_batch": B * world, "seq_len": L, "parallelism": f"batch-sharded x{world}, no collective",
                       "launch": "eager" if args.no_graph else (
                           f"hip-graph replay, {spg} step(s) per graph" + (
                               f", software-pipelined: the loss launch of step i also draws the timesteps and noisy tokens "
                               f"of step i+1 (d3pm_hybrid_loss_then_q_sample), {spg + 1} launches per {spg} steps; every step "
                               "still does one timestep draw, one corruption and one loss fwd+bwd" if pipelined else "")),
                       "two_launches_per_step_ms_per_step": plain_ms,
                       "timed_steps": n_timed, "timed_repeats_of_steps": reps, "timed_seconds": elapsed,
                       "one_step_per_graph_ms_per_step": single_ms,
                       "calls": "training_losses(...).mean().backward()" if args.compat else "training_step(...)",
                       "other_call_shape_ms_per_step": other_ms,
                       "other_call_shape": ("training_step(...)" if args.compat else
                                            f"training_losses(...).mean().backward(), {other_steps} step(s) per graph")},
            "roofline": roof,
            "absorbing": absorbing,
            "reverse_sample": reverse,
            "image_k256": dense,
            "protein_lvb": lvb,
            "large_batch": large,
        }
        if world == 1 and not args.no_cpu_baseline:
            head, lvb_cpu, rev_cpu, img_cpu = cpu_baselines(image_tables=image_tables_cpu)
            out["cpu_baseline"] = head
            if lvb is not None:
                lvb["cpu_baseline"] = lvb_cpu
            if reverse is not None:
                reverse["cpu_baseline"] = rev_cpu
            if dense is not None and img_cpu is not None:
                dense["cpu_baseline"] = img_cpu
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--compat", action="store_true",
                    help="time training_losses(...).mean().backward() instead of the two-launch training_step")
    ap.add_argument("--steps-per-graph", type=int, default=16,
                    help="consecutive steps captured in one HIP graph (a graph launch costs ~19 us of host time: with one "
                         "step per graph the host, not the GPU, bounds the step rate)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="two launches per step (corruption, loss) instead of the loss launch of step i carrying the "
                         "corruption of step i+1")
    ap.add_argument("--roofline-only", action="store_true",
                    help="launch only the sequence roofline.frac is quoted on (the dominant kernel over rotating "
                         "buffers) and print its figure: the program of the committed rocprofv3 kernel trace")
    ap.add_argument("--kind", default="uniform", choices=["uniform", "absorbing"])
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures (reverse sampling, K=256 image configuration, protein-style loss, "
                         "the other transition kind); for counter passes")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if os.environ.get("D3PM_BENCH_SPAWN_ECHO"):
        # launch rehearsal without a GPU: every rank reports the environment it was started with
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            return launch_ranks(args.gpus, argv)
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        path = os.environ["D3PM_BENCH_SPAWN_ECHO"]
        with open(f"{path}.{rec['RANK']}", "w") as f:
            json.dump(rec, f)
        if rec["RANK"] == "0":
            print(json.dumps({"n_gpus": int(rec["WORLD_SIZE"]), "echo": True}), flush=True)
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)          # before any GPU call: the parent never touches the device
    worker(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
