#!/usr/bin/env python3
"""Where the host time of the facade's short calls goes: cProfile of N eager calls (run on the GPU box).

    python tools/host_profile.py [q_sample|training_losses|lvb] [--n 3000]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import d3pm_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="q_sample")
    ap.add_argument("--n", type=int, default=3000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, L, K, T = 256, 512, 32, 500
    d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform",
                                   num_bits=5, transition_bands=None, model_prediction="x_start", model_output="logits",
                                   loss_type="hybrid", hybrid_coeff=0.001, device=dev)
    x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
    t = torch.randint(0, T, (B,), device=dev, dtype=torch.int32)
    logits = torch.randn(B, L, 1, 1, K, device=dev, requires_grad=True)
    if a.what == "q_sample":
        call = lambda: d.q_sample(x0, t)
    elif a.what == "training_losses":
        def call():
            logits.grad = None
            d.training_losses(lambda x, tt: logits, x_start=x0).mean().backward()
    else:
        raise SystemExit("unknown: " + a.what)
    for _ in range(200):
        call()
    torch.cuda.synchronize()
    s = time.perf_counter()
    for _ in range(a.n):
        call()
    torch.cuda.synchronize()
    print(f"{a.what}: {(time.perf_counter() - s) / a.n * 1e6:.2f} us per call (wall clock, eager)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.n):
        call()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
