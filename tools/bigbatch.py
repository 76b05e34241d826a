"""Fused hybrid loss (K = 32) and LVB loss (n_tokens = 24) forward + backward over a range of batch sizes at L = 512,
back-to-back launches between two HIP events: where the per-batch-element launch shape hands over to the chunked one.

    python tools/bigbatch.py [--batches 256,320,384,512,1024,2048] [--L 512] [--msa 0]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import d3pm_amd  # noqa: E402
from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS  # noqa: E402
from d3pm_amd.tables import protein_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,320,384,512,1024,2048")
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--L", type=int, default=512)
    ap.add_argument("--msa", type=int, default=0, help="depth passed to d3pm_lvb_loss (0: single sequences)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    sp = stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L, K, T, lam = args.L, 32, 500, 0.001
    d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform",
                                   num_bits=5, transition_bands=None, model_prediction="x_start", model_output="logits",
                                   loss_type="hybrid", hybrid_coeff=lam, device=dev)
    tb = d.tables()
    tok = d3pm_amd.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=500, device=dev)
    ptab = protein_tables(Qt, Qb)

    def timed(fn):
        until = time.perf_counter() + 0.15
        while time.perf_counter() < until:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps

    for B in [int(b) for b in args.batches.split(",")]:
        x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
        t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
        xt = d.q_sample(x0, t)
        lg = torch.randn(B, L, 1, 1, K, device=dev)
        dl = torch.empty_like(lg)
        ls = torch.empty(B, device=dev)
        wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        us_h = timed(lambda: check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                                        LOSS["hybrid"], lam, 1.0 / B, None, None, None, ptr(ls), ptr(dl),
                                                        None, None, 0, ptr(ws), wsb, sp)))
        del lg, dl
        x0p = torch.randint(0, 22, (B, L), device=dev)
        tp = torch.randint(2, 500, (B,), device=dev)
        xtp = torch.empty_like(x0p)
        check(lib.d3pm_corrupt(ptab.desc(), ptr(x0p), ptr(tp), B, L, None, 1, 0, ptr(xtp), None, sp))
        pr = torch.randn(B, L, 24, device=dev)
        dp = torch.empty_like(pr)
        per = torch.empty(B, dtype=torch.float64, device=dev)
        wsbp = lib.d3pm_lvb_loss_workspace(B, L, 22)
        wsp = torch.empty(max(wsbp // 8, 1), dtype=torch.float64, device=dev)
        us_l = timed(lambda: check(lib.d3pm_lvb_loss(ptab.desc(), ptr(pr), 24, ptr(x0p), ptr(xtp), ptr(tp), None, B, L, 500,
                                                     args.msa, 1.0 / B, ptr(per), ptr(dp), ptr(wsp), wsbp, sp)))
        n = B * L
        print(f"B={B:5d}  hybrid {us_h:7.2f} us  {272 * n / us_h / 8e6:5.3f} of HBM   lvb {us_l:7.2f} us  "
              f"{208 * n / us_l / 8e6:5.3f} of HBM", flush=True)


if __name__ == "__main__":
    main()
