#!/usr/bin/env python3
"""Kernel micro-benchmark: times individual C-ABI entry points back to back on
one stream with HIP events (average over many launches, no host work between
launches beyond the ctypes call) and prints achieved algorithmic bandwidth.

    python tools/kbench.py [--reps 200]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import d3pm_amd
from d3pm_amd._lib import lib, check, ptr, stream_ptr, tok_bytes, NOISE_PHILOX, LOSS


def timeit(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--B", type=int, default=256)
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--bits", type=int, default=5)
    ap.add_argument("--kind", default="uniform")
    ap.add_argument("--same-t", action="store_true", help="one timestep for the whole batch (tables stay in L2)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, K = a.B, a.L, 2 ** a.bits
    T = 1000 if a.kind == "gaussian" else 500
    betas = d3pm_amd.generate_betas(1e-4, 0.02, T, "linear") if a.kind == "gaussian" else d3pm_amd.generate_betas(0, 0, T, "jsd")
    d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type=a.kind,
                                   num_bits=a.bits, transition_bands=None, model_prediction="x_start",
                                   model_output="logits", loss_type="hybrid", hybrid_coeff=0.001, device=dev)
    tb = d.tables()
    x0 = torch.randint(0, K, (B, L), device=dev)
    t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
    if a.same_t:
        t = torch.full_like(t, T // 2)
    logits = torch.randn(B, L, K, device=dev)
    x_t = d.q_sample(x0, t)
    dl = torch.empty_like(logits)
    loss = torch.empty(B, device=dev)
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(wsb // 4, device=dev)
    s = stream_ptr()
    N = B * L

    def f_loss():
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(x_t), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                                   0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl), None, None, 0, ptr(ws), wsb, s))

    mean_acc = torch.zeros((), device=dev)

    def f_loss_mean():      # batch mean accumulated by atomics (accumulator not re-zeroed here: timing only)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(x_t), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                                   0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl), ptr(mean_acc), None, 0, ptr(ws), wsb, s))

    def f_loss_fwd():
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(x_t), 8, ptr(t), 4, B, L, LOSS["hybrid"],
                                   0.001, 1.0 / B, None, None, None, ptr(loss), None, None, None, 0, ptr(ws), wsb, s))

    def f_qs():
        check(lib.d3pm_q_sample(tb.desc(), ptr(x0), 8, ptr(t), 4, B, L, None, NOISE_PHILOX, 1, 0, None, ptr(x_t), s))

    samp = torch.empty_like(x0)

    def f_ps():
        check(lib.d3pm_p_sample(tb.desc(), ptr(logits), K, ptr(x_t), 8, ptr(t), 4, B, L, None, NOISE_PHILOX, 1, 0, None,
                                ptr(samp), None, s))

    def f_copy():
        dl.copy_(logits)

    up = torch.full((B,), 0.5, device=dev)

    def f_noop():
        check(lib.d3pm_scale_rows(ptr(dl), ptr(up), 0.5, B, L * K, s))

    loc = torch.rand(N, device=dev) * 2 - 1
    lsc = torch.randn(N, device=dev) * 0.5
    cen = torch.linspace(-1.0, 1.0, K).to(dev)
    dloc, dlsc = torch.empty_like(loc), torch.empty_like(lsc)

    def f_logistic():
        check(lib.d3pm_logistic_logits(ptr(loc), ptr(lsc), ptr(cen), N, K, 1e-6, ptr(dl), s))

    def f_logistic_bwd():
        check(lib.d3pm_logistic_logits_backward(ptr(loc), ptr(lsc), ptr(cen), ptr(logits), N, K, 1e-6, ptr(dloc), ptr(dlsc), s))

    for name, fn, bytes_per_tok in (("hybrid_loss fwd+bwd", f_loss, 8 * K + 16), ("hybrid_loss fwd+bwd + atomic mean", f_loss_mean, 8 * K + 16),
                                    ("hybrid_loss fwd", f_loss_fwd, 4 * K + 16),
                                    ("q_sample philox", f_qs, 16), ("p_sample philox", f_ps, 4 * K + 16),
                                    ("logistic head fwd", f_logistic, 4 * K + 8), ("logistic head bwd", f_logistic_bwd, 4 * K + 16),
                                    ("torch copy_ (reference stream)", f_copy, 8 * K),
                                    ("scale_rows early exit (launch floor)", f_noop, 0)):
        us = timeit(fn, a.reps)
        print(f"{name:32s} {us:8.2f} us  {N / us * 1e-3:8.2f} Gtok/s  {bytes_per_tok * N / us * 1e-3:8.1f} GB/s "
              f"({bytes_per_tok * N / us * 1e-3 / 8000 * 100:5.1f}% of 8 TB/s)  dense-contraction "
              f"{4 * K * K * N / us * 1e-6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
