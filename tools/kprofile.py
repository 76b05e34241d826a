#!/usr/bin/env python3
"""Every hot kernel of the path at the shapes bench.py uses, launched eagerly through the C ABI (no HIP graphs, no
denoiser, a handful of torch launches for set-up only): the program that the rocprofv3 passes kept under profiles/
run.  Prints each kernel's average duration between two HIP events, back to back on the launch stream, and its
fraction of the 8 TB/s HBM roofline for the algorithmic bytes named in the table below.

    python tools/kprofile.py [--reps 200] [--only hybrid,lvb,...] [--big]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import d3pm_amd  # noqa: E402
from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS  # noqa: E402
from d3pm_amd.tables import protein_tables  # noqa: E402

HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="")
    ap.add_argument("--big", action="store_true",
                    help="also time the two loss kernels at B = 2048 (left out of the profiled runs: the LVB launch would "
                         "share a kernel name with the B = 256 one and mix the per-kernel averages)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    sp = stream_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rows = []

    def timed(name, fn, bytes_per_launch, what):
        if only and name not in only:
            return
        import time
        until = time.perf_counter() + 0.15          # sustained clocks before anything is timed
        while time.perf_counter() < until:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        gbs = bytes_per_launch / us / 1e3
        rows.append({"name": name, "us": us, "GB/s": gbs, "frac_hbm": gbs / HBM_GBS, "what": what})
        print(f"{name:18s} {us:8.2f} us  {gbs:8.1f} GB/s  {gbs / HBM_GBS:5.3f} of HBM peak   {what}", flush=True)

    # ---- image-style, structured tables: B=256, L=512, K=32, T=500
    B, L, K, T, lam = 256, 512, 32, 500, 0.001
    betas = d3pm_amd.generate_betas(0.0, 0.0, T, "jsd")
    x0 = torch.randint(0, K, (B, L, 1, 1), device=dev)
    logits = torch.randn(B, L, 1, 1, K, device=dev)
    dl = torch.empty_like(logits)
    loss = torch.empty(B, device=dev)
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(max(wsb // 4, 1), device=dev)
    t = torch.randint(0, T, (B,), dtype=torch.int32, device=dev)
    for kind in ("uniform", "absorbing"):
        d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=5, transition_bands=None,
                                       model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                       hybrid_coeff=lam, device=dev)
        tb = d.tables()
        xt = d.q_sample(x0, t)

        def hybrid(grad=True):
            check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                       LOSS["hybrid"], lam, 1.0 / B, None, None, None, ptr(loss),
                                       ptr(dl) if grad else None, None, None, 0, ptr(ws), wsb, sp))
        timed(f"hybrid_{kind}", hybrid, (8 * K + 16) * B * L, f"d3pm_hybrid_loss fwd+bwd, {kind} Q, 8K+16 B/token")
        if kind == "uniform":
            timed("hybrid_fwd", lambda: hybrid(False), (4 * K + 16) * B * L, "d3pm_hybrid_loss forward only, 4K+16 B/token")
            timed("q_sample", lambda: d.q_sample(x0, t), 16 * B * L,
                  "q_sample with device noise (facade call), 16 B/token")
            smp = torch.empty(B, L, 1, 1, dtype=torch.int64, device=dev)
            timed("p_sample", lambda: check(lib.d3pm_p_sample(tb.desc(), ptr(logits), K, ptr(xt), 8, ptr(t), 4, B, L, None,
                                                              d3pm_amd._lib.NOISE_PHILOX, 1, 0, None, ptr(smp), None, sp)),
                  (4 * K + 16) * B * L, "d3pm_p_sample (device noise, no probabilities out), 4K+16 B/token")
            t_nx = torch.empty(B, dtype=torch.int32, device=dev)
            xt_nx = torch.empty(B, L, 1, 1, dtype=torch.int64, device=dev)
            timed("hybrid_then_q", lambda: check(lib.d3pm_hybrid_loss_then_q_sample(
                tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L, LOSS["hybrid"], lam, 1.0 / B, None,
                None, None, ptr(loss), ptr(dl), ptr(ws), wsb, ptr(x0), 8, 1, 0, None, ptr(t_nx), ptr(xt_nx), sp)),
                  (8 * K + 32) * B * L, "d3pm_hybrid_loss_then_q_sample: loss fwd+bwd + the next batch's timestep draw "
                  "and corruption in one launch, 8K+32 B/token")
            t_out = torch.empty(B, dtype=torch.int32, device=dev)
            xt_out = torch.empty(B, L, 1, 1, dtype=torch.int64, device=dev)
            timed("q_sample_draw_t", lambda: check(lib.d3pm_q_sample_draw_t(tb.desc(), ptr(x0), 8, B, L, 1, 0, None, None,
                                                                             ptr(t_out), ptr(xt_out), sp)),
                  16 * B * L, "d3pm_q_sample_draw_t (timestep draw + corruption, device noise), 16 B/token")

    # ---- structured tables with wide alphabets (uniform transitions on 6- to 8-bit data): k_hybrid_loss_wide<K / 16>
    for bits in (6, 7, 8):
        KW = 2 ** bits
        if only and f"hybrid_wide{KW}" not in only:
            continue
        dw = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type="uniform", num_bits=bits, transition_bands=None,
                                        model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                        hybrid_coeff=lam, device=dev)
        tbw = dw.tables()
        x0w = torch.randint(0, KW, (B, L, 1, 1), device=dev)
        xtw = dw.q_sample(x0w, t)
        lgw = torch.randn(B, L, 1, 1, KW, device=dev)
        dlw = torch.empty_like(lgw)
        wsbw = lib.d3pm_hybrid_loss_workspace(B, L, KW)
        wsw = torch.empty(max(wsbw // 4, 1), device=dev)
        timed(f"hybrid_wide{KW}", lambda: check(lib.d3pm_hybrid_loss(
            tbw.desc(), ptr(lgw), KW, ptr(x0w), 8, ptr(xtw), 8, ptr(t), 4, B, L, LOSS["hybrid"], lam, 1.0 / B, None, None,
            None, ptr(loss), ptr(dlw), None, None, 0, ptr(wsw), wsbw, sp)),
            (8 * KW + 16) * B * L, f"d3pm_hybrid_loss fwd+bwd, uniform Q, K={KW}, 8K+16 B/token")
        del lgw, dlw

    # ---- protein-style: B=256, L=512, K=22, n_tokens=24; reverse step at 1250 x 263, n_tokens=22
    tok = d3pm_amd.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=500, device=dev)
    ptab = protein_tables(Qt, Qb)
    BP, LP, KP, LDP = 256, 512, 22, 24
    x0p = torch.randint(0, KP, (BP, LP), device=dev)
    tp = torch.randint(2, 500, (BP,), device=dev)
    xtp = torch.empty_like(x0p)
    predp = torch.randn(BP, LP, LDP, device=dev)
    dpp = torch.empty_like(predp)
    perp = torch.empty(BP, dtype=torch.float64, device=dev)
    wsbp = lib.d3pm_lvb_loss_workspace(BP, LP, KP)
    wsp = torch.empty(max(wsbp // 8, 1), dtype=torch.float64, device=dev)

    def corrupt():
        check(lib.d3pm_corrupt(ptab.desc(), ptr(x0p), ptr(tp), BP, LP, None, 1, 0, ptr(xtp), None, sp))

    def lvb():
        check(lib.d3pm_lvb_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 0,
                                1.0 / BP, ptr(perp), ptr(dpp), ptr(wsp), wsbp, sp))
    corrupt()
    timed("corrupt", corrupt, 16 * BP * LP, "d3pm_corrupt (device noise), 16 B/token")
    timed("lvb", lvb, (8 * LDP + 16) * BP * LP, "d3pm_lvb_loss fwd+bwd, n_tokens=24, 8 ld+16 B/token")

    cep = torch.empty(BP, dtype=torch.float64, device=dev)

    def lvb_ce():           # the training loop's loss: (lvb + 0.001 ce) * n_tokens, one launch
        check(lib.d3pm_lvb_hybrid_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 0,
                                       float(LP), 0.001 * LP, ptr(perp), ptr(cep), ptr(dpp), ptr(wsp), wsbp, sp))
    timed("lvb_ce", lvb_ce, (8 * LDP + 16) * BP * LP,
          "d3pm_lvb_hybrid_loss fwd+bwd (LVB + cross-entropy of every token), n_tokens=24, 8 ld+16 B/token")

    def lvb_msa():          # the same tokens taken as MSAs of depth 8 (D3PMLVBLossMSA: msa = depth)
        check(lib.d3pm_lvb_loss(ptab.desc(), ptr(predp), LDP, ptr(x0p), ptr(xtp), ptr(tp), None, BP, LP, 500, 8,
                                1.0 / BP, ptr(perp), ptr(dpp), ptr(wsp), wsbp, sp))
    timed("lvb_msa", lvb_msa, (8 * LDP + 16) * BP * LP, "d3pm_lvb_loss fwd+bwd, MSA variant (depth 8 x 64), n_tokens=24, 8 ld+16 B/token")
    NB, SL = 1250, 263
    for ld in (22, 24):
        xr = torch.randint(0, KP, (NB, SL), device=dev)
        tr = torch.full((NB,), 250, dtype=torch.long, device=dev)
        outr = torch.empty_like(xr)
        frz = torch.randn(NB, SL, ld, device=dev)

        def rev():
            check(lib.d3pm_lvb_reverse_step(ptab.desc(), ptr(frz), ld, ptr(xr), ptr(tr), NB, SL, None, 7, 0, None, 1.0,
                                            0, ptr(outr), sp))
        timed(f"reverse_{ld}", rev, (4 * ld + 16) * NB * SL, f"d3pm_lvb_reverse_step (device noise), n_tokens={ld}, 4 ld+16 B/token")
    # ---- the same two losses at eight times the batch (the global batch of 2048 on one GPU: 1 M tokens, 285 / 218 MB
    # per launch, more than the 256 MiB Infinity Cache holds): what the kernels reach once the launch overhead and
    # the arrival of the first rows are a small part of the launch
    if args.big or only & {"hybrid_b2048", "lvb_b2048", "lvb_ce_b2048"}:
        BB = 2048
        x0b = torch.randint(0, K, (BB, L, 1, 1), device=dev)
        lgb = torch.randn(BB, L, 1, 1, K, device=dev)
        dlb = torch.empty_like(lgb)
        lossb = torch.empty(BB, device=dev)
        tbig = torch.randint(0, T, (BB,), dtype=torch.int32, device=dev)
        wsbb = lib.d3pm_hybrid_loss_workspace(BB, L, K)
        wsb_ = torch.empty(max(wsbb // 4, 1), device=dev)
        du = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type="uniform", num_bits=5, transition_bands=None,
                                        model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                        hybrid_coeff=lam, device=dev)
        tbu = du.tables()
        xtb = du.q_sample(x0b, tbig)
        timed("hybrid_b2048", lambda: check(lib.d3pm_hybrid_loss(
            tbu.desc(), ptr(lgb), K, ptr(x0b), 8, ptr(xtb), 8, ptr(tbig), 4, BB, L, LOSS["hybrid"], lam, 1.0 / BB, None,
            None, None, ptr(lossb), ptr(dlb), None, None, 0, ptr(wsb_), wsbb, sp)),
            (8 * K + 16) * BB * L, "d3pm_hybrid_loss fwd+bwd, uniform Q, B=2048 L=512 K=32")
        del lgb, dlb
        x0q = torch.randint(0, KP, (BB, LP), device=dev)
        tq = torch.randint(2, 500, (BB,), device=dev)
        xtq = torch.empty_like(x0q)
        check(lib.d3pm_corrupt(ptab.desc(), ptr(x0q), ptr(tq), BB, LP, None, 1, 0, ptr(xtq), None, sp))
        predq = torch.randn(BB, LP, LDP, device=dev)
        dpq = torch.empty_like(predq)
        perq = torch.empty(BB, dtype=torch.float64, device=dev)
        wsbq = lib.d3pm_lvb_loss_workspace(BB, LP, KP)
        wsq = torch.empty(max(wsbq // 8, 1), dtype=torch.float64, device=dev)
        timed("lvb_b2048", lambda: check(lib.d3pm_lvb_loss(
            ptab.desc(), ptr(predq), LDP, ptr(x0q), ptr(xtq), ptr(tq), None, BB, LP, 500, 0, 1.0 / BB, ptr(perq),
            ptr(dpq), ptr(wsq), wsbq, sp)),
            (8 * LDP + 16) * BB * LP, "d3pm_lvb_loss fwd+bwd, n_tokens=24, B=2048 L=512")
        ceq = torch.empty(BB, dtype=torch.float64, device=dev)
        timed("lvb_ce_b2048", lambda: check(lib.d3pm_lvb_hybrid_loss(
            ptab.desc(), ptr(predq), LDP, ptr(x0q), ptr(xtq), ptr(tq), None, BB, LP, 500, 0, float(LP), 0.001 * LP,
            ptr(perq), ptr(ceq), ptr(dpq), ptr(wsq), wsbq, sp)),
            (8 * LDP + 16) * BB * LP, "d3pm_lvb_hybrid_loss fwd+bwd, n_tokens=24, B=2048 L=512")
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
