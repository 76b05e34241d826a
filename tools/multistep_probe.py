#!/usr/bin/env python3
"""Step time of the graph-replayed training step against the number of consecutive
steps captured per graph: GPU time per step and host time spent issuing the replays.
One step per graph is bound by the host-side graph launch (~19 us), from about four
steps per graph on the GPU is the bound."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd
B, L, K, T = 256, 512, 32, 500
dev = torch.device('cuda', 0)
d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0., 0., T, 'jsd'), transition_mat_type='uniform', num_bits=5,
                               transition_bands=None, model_prediction='x_start', model_output='logits', loss_type='hybrid',
                               hybrid_coeff=0.001, device=dev)
x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
logits = torch.randn(B, L, 1, 1, K, device=dev, requires_grad=True)
fn = lambda x, t: logits
counter = torch.zeros(1, dtype=torch.int64, device=dev)
d.tables()
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
for n in (1, 2, 4, 8, 16):
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            d.training_step(fn, x_start=x0, rng=(1, counter))
    torch.cuda.current_stream(dev).wait_stream(side)
    logits.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            logits.grad = None
            out = d.training_step(fn, x_start=x0, rng=(1, counter))
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    reps = 1600 // n
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps/graph {n:2d}: {1e6*(t2-t0)/(reps*n):7.2f} us/step  (host issue {1e6*(t1-t0)/(reps*n):6.2f} us/step)", flush=True)
