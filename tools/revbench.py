#!/usr/bin/env python3
"""Duration of one protein-style reverse step (device noise), back-to-back launches between HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd
from d3pm_amd.protein_generation.generate import reverse_step
from d3pm_amd.tables import protein_tables

dev = torch.device("cuda", 0)
B, L, K = 1024, 263, 22
for ld in (22, 24):
    tok = d3pm_amd.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=500, device=dev)
    tb = protein_tables(Qt, Qb)
    logits = torch.randn(B, L, ld, device=dev)
    x = torch.randint(0, K, (B, L), device=dev)
    tt = torch.full((B,), 250, dtype=torch.long, device=dev)
    c = torch.zeros(1, dtype=torch.int64, device=dev)
    g = torch.cuda.CUDAGraph()
    for _ in range(3):
        reverse_step(tb, logits, x, tt, rng=(1, 0, c))
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(50):
            y = reverse_step(tb, logits, x, tt, rng=(1, 4 * i, c))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 500
    byts = B * L * (4 * ld + 16)
    print(f"reverse step B={B} L={L} ld={ld}: {us:6.2f} us per step  {byts / us / 1e3:7.1f} GB/s ({byts / us / 1e3 / 80:.1f}% of 8 TB/s)")
