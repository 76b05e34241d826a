#!/usr/bin/env python3
"""Duration of the protein-style LVB loss (forward + gradient) and of the batched forward corruption,
back-to-back launches through the C ABI between HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd
from d3pm_amd._lib import lib, check, ptr, stream_ptr
from d3pm_amd.tables import protein_tables

dev = torch.device("cuda", 0)
K = 22


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


tok = d3pm_amd.Tokenizer()
Qb, Qt = tok.q_uniform_schedule(num_steps=500, device=dev)
tb = protein_tables(Qt, Qb)
for B, L, ld in ((256, 512, 24), (256, 263, 24), (256, 512, 22), (2048, 512, 24)):
    x0 = torch.randint(0, K, (B, L), device=dev)
    t = torch.randint(2, 500, (B,), device=dev)
    xt = torch.empty_like(x0)
    check(lib.d3pm_corrupt(tb.desc(), ptr(x0), ptr(t), B, L, None, 1, 0, ptr(xt), None, stream_ptr()))
    pred = torch.randn(B, L, ld, device=dev)
    dp = torch.empty_like(pred)
    per = torch.empty(B, dtype=torch.float64, device=dev)
    wsb = lib.d3pm_lvb_loss_workspace(B, L, K)
    ws = torch.empty(wsb // 8, dtype=torch.float64, device=dev)
    sp = stream_ptr()
    f_loss = lambda: check(lib.d3pm_lvb_loss(tb.desc(), ptr(pred), ld, ptr(x0), ptr(xt), ptr(t), None, B, L, 500, 0,
                                             1.0 / B, ptr(per), ptr(dp), ptr(ws), wsb, sp))
    ce = torch.empty(B, dtype=torch.float64, device=dev)
    f_hyb = lambda: check(lib.d3pm_lvb_hybrid_loss(tb.desc(), ptr(pred), ld, ptr(x0), ptr(xt), ptr(t), None, B, L, 500, 0,
                                                   float(L), 0.001 * L, ptr(per), ptr(ce), ptr(dp), ptr(ws), wsb, sp))
    f_fwd = lambda: check(lib.d3pm_lvb_loss(tb.desc(), ptr(pred), ld, ptr(x0), ptr(xt), ptr(t), None, B, L, 500, 0,
                                            1.0 / B, ptr(per), None, ptr(ws), wsb, sp))
    f_cor = lambda: check(lib.d3pm_corrupt(tb.desc(), ptr(x0), ptr(t), B, L, None, 1, 0, ptr(xt), None, sp))
    N = B * L
    for name, f, byts in (("lvb loss fwd+bwd", f_loss, N * (8 * ld + 16)), ("lvb + ce fwd+bwd", f_hyb, N * (8 * ld + 16)),
                          ("lvb loss fwd", f_fwd, N * (4 * ld + 16)),
                          ("corrupt (device noise)", f_cor, N * 16)):
        us = timeit(f)
        print(f"B={B} L={L} ld={ld}  {name:24s} {us:8.2f} us  {N / us / 1e3:7.2f} Gtok/s  {byts / us / 1e3:7.1f} GB/s "
              f"({byts / us / 1e3 / 80:.1f}% of 8 TB/s)")
