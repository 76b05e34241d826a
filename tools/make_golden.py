#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference.

Runs only in the development container, where the upstream reference tree is
mounted read-only (default /root/reference, override with D3PM_REFERENCE).
The two reference sub-projects both have a top-level ``utils.py``, so each is
imported in its own child process (``--part image`` / ``--part protein``).
Only inputs and outputs are written (small ``.npz`` files); no reference source
is copied.  The GPU box never runs this script and never sees the reference.

    python tools/make_golden.py            # regenerate everything
"""
from __future__ import annotations

import argparse
import contextlib
import io
import tempfile
import hashlib
import os
import subprocess
import sys
import types

import numpy as np

REF = os.environ.get("D3PM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
FASTA = os.path.join(REF, "protein_generation", "generated_sequences", "generated_samples_string.fasta")
ALPHABET = "ACDEFGHIKLMNPQRSTVWYX-#!"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fasta_tokens(n: int) -> np.ndarray:
    seqs = [ln.strip() for ln in open(FASTA) if not ln.startswith(">")][:n]
    lut = {c: i for i, c in enumerate(ALPHABET)}
    return np.array([[lut[c] for c in s] for s in seqs], dtype=np.int64)


ONLY = None          # --only: names of the fixtures to (re)write; the others are computed and left as they are


def save(name: str, **arrs):
    if ONLY is not None and name not in ONLY:
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrs)} arrays")


SLICE_T = lambda T: sorted({0, 1, 2, T // 2, T - 2, T - 1})


# ---------------------------------------------------------------------------
# image-style reference (d3pm/)
# ---------------------------------------------------------------------------
def part_image():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "d3pm"))
    import torch
    import diffusion_discrete as dd   # the reference
    import utils as ru                # the reference's d3pm/utils.py

    torch.set_num_threads(8)
    import torch as _t
    info = np.array([_t.__version__, np.__version__])

    def table_arrays(d, prefix=""):
        T = d.num_timesteps
        ts = SLICE_T(T)
        qm = d.q_mats.numpy()
        qt = d.transpose_q_onestep_mats.contiguous().numpy()
        return {prefix + "slice_t": np.array(ts), prefix + "q_mats_slices": qm[ts],
                prefix + "qT_onestep_slices": qt[ts],
                prefix + "q_mats_sha256": np.array(sha(qm)), prefix + "qT_onestep_sha256": np.array(sha(qt))}

    def downstream(d, x0, t, seed, frozen_scale=1.0):
        """Every per-step method on explicit noise and frozen denoiser logits."""
        g = torch.Generator().manual_seed(seed)
        K = d.num_pixel_vals
        noise = torch.rand(x0.shape + (K,), generator=g)
        logits = (torch.randn(x0.shape + (K,), generator=g) * frozen_scale).requires_grad_(True)
        noise2 = torch.rand(x0.shape + (K,), generator=g)
        w = torch.rand(x0.shape[0], generator=g) + 0.5
        model_fn = lambda x, tt: logits
        out = {}
        out["x0"], out["t"], out["noise"], out["logits"] = x0.numpy(), t.numpy(), noise.numpy(), logits.detach().numpy()
        out["noise2"], out["w"] = noise2.numpy(), w.numpy()
        out["q_probs"] = d.q_probs(x0, t).numpy()
        x_t = d.q_sample(x0, t, noise)
        out["x_t"] = x_t.numpy()
        lg = torch.log(d.q_probs(x0, t) + d.eps) + (-torch.log(-torch.log(torch.clip(noise, min=torch.finfo(noise.dtype).tiny, max=1.))))
        top2 = torch.topk(lg, 2, dim=-1).values
        out["q_sample_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        out["post_true"] = d.q_posterior_logits(x0, x_t, t, x_start_logits=False).numpy()
        out["post_model"] = d.q_posterior_logits(logits.detach(), x_t, t, x_start_logits=True).numpy()
        ml, px0 = d.p_logits(model_fn, x=x_t, t=t)
        out["p_logits"] = ml.detach().numpy()
        vb, _ = d.vb_terms_bpd(model_fn, x_start=x0, x_t=x_t, t=t)
        out["vb"] = vb.detach().numpy()
        out["ce"] = d.cross_entropy_start(x0, logits).detach().numpy()
        # training_losses draws its own noise/t; replay its body on the recorded x_t
        for lt in ("hybrid", "kl", "cross_entropy_x_start"):
            d.loss_type = lt
            if lt == "kl":
                losses, _ = d.vb_terms_bpd(model_fn=model_fn, x_start=x0, x_t=x_t, t=t)
            elif lt == "cross_entropy_x_start":
                _, pl = d.p_logits(model_fn, x=x_t, t=t)
                losses = d.cross_entropy_start(x_start=x0, pred_x_start_logits=pl)
            else:
                vbl, pl = d.vb_terms_bpd(model_fn=model_fn, x_start=x0, x_t=x_t, t=t)
                losses = vbl + d.hybrid_coeff * d.cross_entropy_start(x_start=x0, pred_x_start_logits=pl)
            out["losses_" + lt] = losses.detach().numpy()
            (gr,) = torch.autograd.grad((losses * w).sum(), logits)
            out["grad_w_" + lt] = gr.numpy()
        d.loss_type = "hybrid"
        sample, probs = d.p_sample(model_fn, x=x_t, t=t, noise=noise2)
        out["p_sample"], out["p_sample_probs"] = sample.numpy(), probs.numpy()
        mask = (t != 0).float().view(-1, 1, 1, 1, 1)
        lg = ml.detach() + mask * (-torch.log(-torch.log(torch.clip(noise2, min=torch.finfo(noise2.dtype).tiny, max=1.))))
        top2 = torch.topk(lg, 2, dim=-1).values
        out["p_sample_margin"] = (top2[..., 0] - top2[..., 1]).numpy()
        out["prior_bpd"] = d.prior_bpd(x0).numpy() if d.transition_mat_type != "absorbing" else np.zeros(0)
        return out

    def make(kind, num_bits, betas, bands=None, lam=0.001):
        return dd.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=num_bits,
                                    transition_bands=bands, model_prediction="x_start", model_output="logits",
                                    loss_type="hybrid", hybrid_coeff=lam, device="cpu")

    toks = fasta_tokens(6)[:, :96]
    pos = np.arange(toks.shape[1])[None, :]

    # --- uniform, K=32, T=500 (jsd) : the protein-synthetic configuration -----
    betas = dd.generate_betas(0.0, 0.0, 500, "jsd")
    d = make("uniform", 5, betas)
    x0 = torch.tensor(((toks * 3 + pos) % 32).astype(np.int32)).reshape(6, 96, 1, 1)
    t = torch.tensor([0, 1, 2, 250, 498, 499], dtype=torch.int32)
    arrs = {"betas": betas.numpy(), "torch_numpy_versions": info, **table_arrays(d), **downstream(d, x0, t, 11)}
    save("image_uniform_k32", **arrs)

    # --- uniform, cosine betas: tables only --------------------------------
    betas = dd.generate_betas(0.0, 0.0, 500, "cosine")
    d = make("uniform", 5, betas)
    save("image_uniform_k32_cosine_tables", betas=betas.numpy(), **table_arrays(d))

    # --- absorbing, K=32, T=500: builder crashes in the reference (torch.full(shape=)),
    #     so inject tables built from its evident formula and run every downstream method.
    betas = dd.generate_betas(0.0, 0.0, 500, "jsd")
    d = make("uniform", 5, betas)
    K = 32
    steps = []
    for tt in range(500):
        m = np.diag(np.full(K, 1.0 - betas[tt].item()))
        m[:, K // 2] += betas[tt].item()
        steps.append(m)
    cur = steps[0]
    chain = [cur]
    for tt in range(1, 500):
        cur = np.tensordot(cur, steps[tt], axes=[[1], [0]])
        chain.append(cur)
    d.q_mats = torch.tensor(np.stack(chain))
    d.transpose_q_onestep_mats = torch.tensor(np.stack(steps)).permute(0, 2, 1)
    d.transition_mat_type = "absorbing"
    arrs = {"betas": betas.numpy(), **table_arrays(d), **downstream(d, x0, t, 12)}
    save("image_absorbing_k32", **arrs)

    # --- gaussian, K=256, T=1000, linear 1e-4 -> 0.02: the image configuration ---
    betas = dd.generate_betas(1e-4, 0.02, 1000, "linear")
    d = make("gaussian", 8, betas)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randint(0, 256, (4, 6, 6, 1), generator=g, dtype=torch.int32)
    t = torch.tensor([0, 1, 500, 999], dtype=torch.int32)
    arrs = {"betas": betas.numpy(), **table_arrays(d), **downstream(d, x0, t, 13, frozen_scale=2.0)}
    save("image_gaussian_k256", **arrs)

    # --- small full tables: gaussian (full band and 5 bands), banded uniform ------
    for name, kind, bands in (("gaussian_k16_full", "gaussian", None), ("gaussian_k16_b5", "gaussian", 5),
                              ("uniform_k16_b3", "uniform", 3)):
        betas = dd.generate_betas(0.002, 0.3, 40, "linear")
        d = make(kind, 4, betas, bands=bands)
        g = torch.Generator().manual_seed(6)
        x0 = torch.randint(0, 16, (4, 5, 3, 2), generator=g, dtype=torch.int64)
        t = torch.tensor([0, 1, 20, 39], dtype=torch.int64)
        arrs = {"betas": betas.numpy(), "q_mats": d.q_mats.numpy(),
                "qT_onestep": d.transpose_q_onestep_mats.contiguous().numpy(),
                "bands": np.array(-1 if bands is None else bands), **downstream(d, x0, t, 14)}
        save("image_" + name, **arrs)

    # --- schedules and helpers ------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    a = -torch.rand(64, generator=g) * 3
    b = a - torch.rand(64, generator=g) * 2 - 1e-3
    loc = torch.rand(3, 4, 4, 1, generator=g) * 2 - 1
    log_scale = torch.randn(3, 4, 4, 1, generator=g) * 0.5
    d = make("uniform", 4, dd.generate_betas(0.0, 0.0, 20, "jsd"))
    d.device = "cpu"
    # gradient of the logistic head as autograd derives it, under a fixed weighting of the logits
    loc_g, ls_g = loc.clone().requires_grad_(True), log_scale.clone().requires_grad_(True)
    w_log = torch.randn(3, 4, 4, 1, 16, generator=g)
    (d._get_logits_from_logistic_pars(loc_g, ls_g) * w_log).sum().backward()
    save("image_misc",
         betas_linear=dd.generate_betas(1e-4, 0.02, 1000, "linear").numpy(),
         betas_cosine=dd.generate_betas(0.0, 0.0, 500, "cosine").numpy(),
         betas_jsd=dd.generate_betas(0.0, 0.0, 500, "jsd").numpy(),
         lme_a=a.numpy(), lme_b=b.numpy(), lme_out=ru.log_min_exp(a, b).numpy(),
         logistic_loc=loc.numpy(), logistic_log_scale=log_scale.numpy(),
         logistic_logits=d._get_logits_from_logistic_pars(loc, log_scale).numpy(),
         logistic_w=w_log.numpy(), logistic_grad_loc=loc_g.grad.numpy(), logistic_grad_log_scale=ls_g.grad.numpy())

    # --- calc_bpd_loop with the logistic_pars head: gaussian K=16, T=40, frozen (loc, log_scale) -----------------
    betas = dd.generate_betas(0.002, 0.3, 40, "linear")
    d = dd.DiscreteDiffusion(betas=betas, transition_mat_type="gaussian", num_bits=4, transition_bands=None,
                             model_prediction="x_start", model_output="logistic_pars", loss_type="hybrid",
                             hybrid_coeff=0.001, device="cpu")
    g = torch.Generator().manual_seed(8)
    x0 = torch.randint(0, 16, (3, 5, 2, 1), generator=g, dtype=torch.int64)
    loc = torch.rand(3, 5, 2, 1, generator=g) * 2 - 1
    log_scale = torch.randn(3, 5, 2, 1, generator=g) * 0.5
    torch.manual_seed(4242)
    res = d.calc_bpd_loop(lambda x, tt: (loc, log_scale), x_start=x0)
    torch.manual_seed(4242)      # the draws of :611, in order
    noises = torch.stack([torch.rand(x0.shape + (16,)) for _ in range(40)])
    save("image_bpd_loop", betas=betas.numpy(), x0=x0.numpy(), loc=loc.numpy(), log_scale=log_scale.numpy(),
         noises=noises.numpy(), total=res["total"].numpy(), vbterms=res["vbterms"].numpy(), prior=res["prior"].numpy())


# ---------------------------------------------------------------------------
# protein-style reference (protein_generation/)
# ---------------------------------------------------------------------------
def part_protein():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "protein_generation"))
    # generate.py imports dataset.py, which parses a Colab path at import time.
    stub = types.ModuleType("dataset")
    stub.ProteinSequenceDatasetMSA = stub.CollaterMSA = object
    sys.modules["dataset"] = stub
    import torch
    import utils as pu
    import losses as pl
    import generate as pg

    torch.set_num_threads(8)
    tok = pu.Tokenizer()
    K = tok.K
    assert K == 22

    # --- schedules and tables --------------------------------------------------------
    Qb, Qt = tok.q_uniform_schedule(num_steps=500, schedule="sohl-dickstein")
    ts = SLICE_T(500)
    save("protein_tables_k22",
         betas_sohl=pu._beta_schedule(500, schedule="sohl-dickstein").numpy(),
         betas_linear=pu._beta_schedule(500, schedule="linear").numpy(),
         betas_cosine=pu._beta_schedule(500, schedule="cosine").numpy(),
         slice_t=np.array(ts), Q_bar_slices=Qb.numpy()[ts], Q_t_slices=Qt.numpy()[ts],
         Q_bar_sha256=np.array(sha(Qb.numpy())), Q_t_sha256=np.array(sha(Qt.numpy())))
    Qb_lin, Qt_lin = tok.q_uniform_schedule(num_steps=60, schedule="linear")
    save("protein_tables_k22_linear60", Q_bar=Qb_lin.numpy(), Q_t=Qt_lin.numpy())

    # --- forward corruption with the multinomial draw recorded -----------------------
    toks = torch.tensor(fasta_tokens(8))
    B, L = toks.shape
    tsteps = [1, 2, 250, 499, 500, 37, 1, 420]
    x_t, q_x, E = [], [], []
    for i in range(B):
        tt = min(tsteps[i], 499)                      # CollaterV2: t==num_steps corrupts with Q_bar[t-1]
        oh = tok.one_hot(toks[i])
        torch.manual_seed(100 + i)
        s, p = pu.sample_transition_matrix(oh, Qb[tt])
        torch.manual_seed(100 + i)
        e = torch.empty_like(p).exponential_(1)
        assert torch.equal(torch.argmax(p / e, dim=1), s), "multinomial != argmax(p/E) on this torch build"
        x_t.append(s); q_x.append(p); E.append(e)
    x_t, q_x, E = torch.stack(x_t), torch.stack(q_x), torch.stack(E)

    # --- LVB loss, single-sequence, n_tokens = 22 and 24 --------------------------------
    src_oh = torch.stack([tok.one_hot(s) for s in x_t])
    tgt_oh = torch.stack([tok.one_hot(s) for s in toks])
    timestep = torch.tensor(tsteps, dtype=torch.long)
    arrs = dict(x0=toks.numpy(), x_t=x_t.numpy(), q=q_x.numpy(), expo=E.numpy(), timestep=timestep.numpy(),
                torch_version=np.array(torch.__version__))
    for n_tok in (22, 24):
        g = torch.Generator().manual_seed(200 + n_tok)
        pred = (torch.randn(B, L, n_tok, generator=g) * 2).requires_grad_(True)
        lf = pl.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)
        per = lf(src_oh, q_x, pred, toks, tgt_oh, timestep, Qt, Qb)
        lf2 = pl.D3PMLVBLoss(tmax=500, tokenizer=tok)
        lvb = lf2(src_oh, q_x, pred, toks, tgt_oh, timestep, Qt, Qb)
        (gr,) = torch.autograd.grad(lvb, pred)
        ce = torch.nn.CrossEntropyLoss(reduction="mean")(pred.reshape(-1, n_tok), toks.reshape(-1))
        arrs.update({f"pred{n_tok}": pred.detach().numpy(), f"per_seq{n_tok}": per.detach().numpy(),
                     f"lvb{n_tok}": lvb.detach().numpy(), f"grad{n_tok}": gr.numpy(), f"ce{n_tok}": ce.detach().numpy()})
    save("protein_lvb_blat", **arrs)

    # --- the training loop's loss around the module (Protein Generation Notebook cell-12) --------------
    # loss = (lvb + _lambda * ce) * n_tokens, both terms cast to float32; inputs are those of protein_lvb_blat
    hyb = {}
    for n_tok in (22, 24):
        for lam in (0.001, 0.5):
            pred = torch.tensor(arrs[f"pred{n_tok}"]).requires_grad_(True)
            lf = pl.D3PMLVBLoss(tmax=500, tokenizer=tok)
            lvb_loss = lf(src_oh, q_x, pred, toks, tgt_oh, timestep, Qt, Qb)
            ce_loss = torch.nn.CrossEntropyLoss(weight=None, reduction="mean")(pred.reshape(-1, pred.shape[-1]), toks.reshape(-1))
            n_tokens = x_t.shape[0] * x_t.shape[1]
            loss = (lvb_loss.to(torch.float32) + (lam * ce_loss.to(torch.float32))) * n_tokens
            (gr,) = torch.autograd.grad(loss, pred)
            hyb.update({f"loss{n_tok}_lam{lam}": loss.detach().numpy(), f"grad{n_tok}_lam{lam}": gr.numpy()})

    # --- LVB loss, MSA variant -----------------------------------------------------------
    g = torch.Generator().manual_seed(300)
    Bm, D, Lm = 4, 3, 40
    x0m = torch.randint(0, K, (Bm, D, Lm), generator=g)
    tm = torch.tensor([1, 7, 333, 500], dtype=torch.long)
    xtm, qm = [], []
    for i in range(Bm):
        oh = tok.one_hot(x0m[i]).flatten(0, 1)
        p = torch.mm(oh, Qb[min(int(tm[i]), 499)])
        e = torch.empty_like(p).exponential_(1, generator=g)
        xtm.append(torch.argmax(p / e, dim=1).reshape(D, Lm)); qm.append(p.reshape(D, Lm, K))
    xtm, qm = torch.stack(xtm), torch.stack(qm)
    predm = (torch.randn(Bm, D, Lm, 24, generator=g) * 2).requires_grad_(True)
    lfm = pl.D3PMLVBLossMSA(tmax=500, tokenizer=tok)
    lvbm = lfm(tok.one_hot(xtm), qm, predm, x0m, tok.one_hot(x0m), tm, Qt, Qb)
    (grm,) = torch.autograd.grad(lvbm, predm)
    save("protein_lvb_msa", x0=x0m.numpy(), x_t=xtm.numpy(), q=qm.numpy(), timestep=tm.numpy(),
         pred=predm.detach().numpy(), lvb=lvbm.detach().numpy(), grad=grm.numpy())
    # cell-13: the same loop around the MSA module (n_tokens = src.shape[0] * src.shape[1] = B * D there)
    for lam in (0.001, 0.5):
        pm = predm.detach().clone().requires_grad_(True)
        lvb_loss = lfm(tok.one_hot(xtm), qm, pm, x0m, tok.one_hot(x0m), tm, Qt, Qb)
        ce_loss = torch.nn.CrossEntropyLoss(weight=None, reduction="mean")(pm.reshape(-1, pm.shape[-1]), x0m.reshape(-1))
        n_tokens = xtm.shape[0] * xtm.shape[1]
        loss = (lvb_loss.to(torch.float32) + (lam * ce_loss.to(torch.float32))) * n_tokens
        (gr,) = torch.autograd.grad(loss, pm)
        hyb.update({f"msa_loss_lam{lam}": loss.detach().numpy(), f"msa_grad_lam{lam}": gr.numpy(),
                    "msa_ce": ce_loss.detach().numpy()})
    save("protein_driver_loss", **hyb)

    # --- full reverse sampling run with a toy denoiser, draws replayed -------------------
    T, Bg, Lg = 20, 3, 48
    Qb20, Qt20 = tok.q_uniform_schedule(num_steps=T, schedule="sohl-dickstein")
    g = torch.Generator().manual_seed(400)
    emb = torch.randn(K, 24, generator=g) * 1.5
    tvec = torch.randn(T + 1, 24, generator=g) * 0.3

    class Toy(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, :]

    torch.manual_seed(401)
    sample, strings = pg.generate_d3pm(Toy(), tok, Qt20, Qb20, T, Lg, "cpu", batch_size=Bg)
    torch.manual_seed(401)
    x_init = torch.randint(0, K, (Bg, Lg))
    Es = torch.stack([torch.stack([torch.empty(Lg, K, dtype=torch.float64).exponential_(1) for _ in range(Bg)])
                      for _ in range(T - 1)])
    save("protein_generate", Q_bar=Qb20.numpy(), Q_t=Qt20.numpy(), emb=emb.numpy(), tvec=tvec.numpy(),
         x_init=x_init.numpy(), expo=Es.numpy(), sample=sample.numpy(), strings=np.array(strings))


    # --- MSA reverse sampling (no query pinning: the query comes from a data file) ------
    Tm, Bs, Ds, Ls = 12, 2, 3, 16
    Qb12, Qt12 = tok.q_uniform_schedule(num_steps=Tm, schedule="sohl-dickstein")

    class ToyMSA(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, None, :]

    penalty = 0.75
    torch.manual_seed(501)
    with contextlib.redirect_stdout(io.StringIO()):
        sample_m, strings_m = pg.generate_msa_d3pm(ToyMSA(), Bs, Ds, Ls, "cpu", timesteps=Tm, penalty_value=penalty,
                                                   Q_bar=Qb12, Q=Qt12, tokenizer=tok, start_query=False)
    torch.manual_seed(501)
    x_init_m = torch.randint(0, K, (Bs, Ds, Ls))
    Ea, Eq = [], []
    for _ in range(Tm - 1):
        ea, eq = [], []
        for _ in range(Bs):
            ea.append(torch.empty(Ds * Ls, K, dtype=torch.float64).exponential_(1))
            eq.append(torch.empty(Ls, K - 1, dtype=torch.float64).exponential_(1))
        Ea.append(torch.stack(ea)); Eq.append(torch.stack(eq))
    save("protein_generate_msa", Q_bar=Qb12.numpy(), Q_t=Qt12.numpy(), emb=emb.numpy(), tvec=tvec.numpy(),
         x_init=x_init_m.numpy(), expo_all=torch.stack(Ea).numpy(), expo_query=torch.stack(Eq).numpy(),
         penalty=np.array(penalty), sample=sample_m.numpy(), strings=np.array([s[0] for s in strings_m]))

    # --- CollaterV2 (utils.py:240-293) with its draws recorded -----------------------------
    seqs = [tok.untokenize(r) for r in fasta_tokens(4)]
    col = pu.CollaterV2(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb)
    arrs = dict(tokens=fasta_tokens(4))
    for tcase in (1, 37, 500):
        torch.manual_seed(600 + tcase)
        src, src_oh, tsv, tgt, tgt_oh, _, _, qx = col(seqs, tcase)
        torch.manual_seed(600 + tcase)
        Ec = torch.stack([torch.empty(len(seqs[0]), K, dtype=torch.float64).exponential_(1) for _ in seqs])
        arrs.update({f"src_t{tcase}": src.numpy(), f"timesteps_t{tcase}": tsv.numpy(), f"q_t{tcase}": qx.numpy(),
                     f"expo_t{tcase}": Ec.numpy()})
    save("protein_collate_v2", **arrs)

    # --- Collater / CollaterMSA (dataset.py:196-300) ------------------------------------------
    # dataset.py parses its training alignment at import (a Colab path, :19, :119) and writes a FASTA beside the
    # working directory: the module is executed as it is, with that one path served from a temporary alignment built
    # from the generated sequences, in a temporary working directory.
    import builtins
    import importlib.util
    tmpd = tempfile.mkdtemp()
    a2m = os.path.join(tmpd, "train.a2m")
    rows16 = fasta_tokens(16)
    with open(a2m, "w") as f:
        for i, r in enumerate(rows16):
            f.write(f">SEQ_{i}/1-{len(r)}\n{tok.untokenize(r)}\n")
    real_open = builtins.open

    def open_redirect(path, *a, **k):
        if isinstance(path, str) and path.startswith("/content/drive/"):
            path = a2m
        return real_open(path, *a, **k)

    cwd = os.getcwd()
    os.chdir(tmpd)
    builtins.open = open_redirect
    try:
        spec = importlib.util.spec_from_file_location("ref_dataset", os.path.join(REF, "protein_generation", "dataset.py"))
        rds = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rds)
    finally:
        builtins.open = real_open
        os.chdir(cwd)
    seqs8 = [tok.untokenize(r) for r in rows16[:8]]
    col = rds.Collater(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb)
    np.random.seed(21); torch.manual_seed(901)
    src, src_oh, tsv, tgt, tgt_oh, Qr, Qbr, qx, names = col([(s_, f"n{i}") for i, s_ in enumerate(seqs8)])
    assert Qr is Qt and Qbr is Qb
    torch.manual_seed(901)
    Ec = torch.stack([torch.empty(len(seqs8[0]), K, dtype=torch.float64).exponential_(1) for _ in seqs8])
    carr = dict(tokens=rows16, np_seed=np.array(21), src=src.numpy(), timesteps=tsv.numpy(), q_rows=qx.numpy()[:, :4],
                expo=Ec.numpy(), names=np.array(names), src_one_hot_sha256=np.array(sha(src_oh.numpy())),
                tgt_one_hot_sha256=np.array(sha(tgt_oh.numpy())), tokenized=tgt.numpy())
    # MSAs: 3 alignments of 4 rows, first 60 columns
    msas = [[tok.untokenize(r[:60]) for r in rows16[4 * m:4 * m + 4]] for m in range(3)]
    colm = rds.CollaterMSA(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb, num_seqs=4)
    np.random.seed(22); torch.manual_seed(902)
    srcm, srcm_oh, tsm, tgtm, tgtm_oh, _, _, qxm = colm(msas)
    torch.manual_seed(902)
    Em = torch.stack([torch.empty(4 * 60, K, dtype=torch.float64).exponential_(1) for _ in msas])
    carr.update(msa_np_seed=np.array(22), msa_src=srcm.numpy(), msa_timesteps=tsm.numpy(), msa_q_shape=np.array(qxm.shape),
                msa_q_rows=qxm.numpy()[:, :, :3], msa_expo=Em.numpy(), msa_tokenized=tgtm.numpy(),
                msa_src_one_hot_sha256=np.array(sha(srcm_oh.numpy())))
    save("protein_collaters", **carr)

    # --- custom_mutant_matrix (utils.py:105-236) on a synthetic mutant list ----------------
    Lf, Tf = 24, 30
    Qb30, Qt30 = tok.q_uniform_schedule(num_steps=Tf, schedule="sohl-dickstein")
    rng = np.random.RandomState(7)
    wt = "".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), Lf))
    focus_seq = wt[:5] + wt[5:8].lower() + wt[8:]               # three unmodelled (lower-case) columns
    focus_trimmed = [c for c in focus_seq if c == c.upper()]
    start = 24
    lines = ["mutant,effect"]
    cols = [i for i, c in enumerate(focus_seq) if c == c.upper()]
    for k in range(9):
        i = cols[int(rng.randint(len(cols)))]
        mut = rng.choice([a for a in "ACDEFGHIKLMNPQRSTVWY" if a != focus_seq[i]])
        m = f"{focus_seq[i]}{start + i}{mut}"
        if k % 4 == 3:
            i2 = cols[(cols.index(i) + 3) % len(cols)]
            mut2 = rng.choice([a for a in "ACDEFGHIKLMNPQRSTVWY" if a != focus_seq[i2]])
            m += f":{focus_seq[i2]}{start + i2}{mut2}"
        lines.append(m + ",0.0")
    lines.append(f"{focus_seq[5].upper()}{start + 5}A,0.0")       # a lower-case column: not a valid mutant
    csv_path = os.path.join(tempfile.mkdtemp(), "mutants.csv")
    with open(csv_path, "w") as f:
        f.write("\n".join(lines) + "\n")

    class ToyT(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, :]

    n_iter, mb = 2, 4
    lf_eval = pl.D3PMLVBLoss(tmax=Tf, tokenizer=tok, evaluation=True)
    np.random.seed(11); torch.manual_seed(701)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        desc, delta = pu.custom_mutant_matrix(csv_path, ToyT(), "cpu", focus_seq, f"TOY/{start}-{start + Lf - 1}",
                                              focus_trimmed, tok, lf_eval, None, 0.0, Tf, Qt30, Qb30,
                                              N_pred_iterations=n_iter, minibatch_size=mb)
    # replay numpy's shuffle / randint and torch's exponential draws in call order
    n_seq = len(desc) + 1
    np.random.seed(11); torch.manual_seed(701)
    orders, tmid, Em = [], [], []
    order = np.arange(n_seq)
    for i in range(n_iter):
        np.random.shuffle(order)
        orders.append(order.copy())
        for j in range(0, n_seq, mb):
            nb = len(order[j:j + mb])
            for h in range(3):
                if h == 1:
                    tmid.append(np.random.randint(2, Tf))
                Em.append(torch.stack([torch.empty(len(focus_trimmed), K, dtype=torch.float64).exponential_(1)
                                       for _ in range(nb)]).numpy())
    save("protein_mutant_scores", csv=np.array("\n".join(lines) + "\n"), focus_seq=np.array(focus_seq),
         focus_name=np.array(f"TOY/{start}-{start + Lf - 1}"), steps=np.array(Tf), n_iter=np.array(n_iter),
         minibatch=np.array(mb), Q_bar=Qb30.numpy(), Q_t=Qt30.numpy(), emb=emb.numpy(), tvec=tvec.numpy(),
         descriptors=np.array(desc), delta_elbos=np.asarray(delta), orders=np.stack(orders), t_mid=np.array(tmid),
         expo=np.array(Em, dtype=object) if False else np.concatenate([e.reshape(-1) for e in Em]),
         expo_rows=np.array([e.shape[0] for e in Em]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", choices=["image", "protein"], default=None)
    ap.add_argument("--only", default=None, help="comma-separated fixture names to write (default: all)")
    a = ap.parse_args()
    global ONLY
    ONLY = set(a.only.split(",")) if a.only else None
    if a.part == "image":
        part_image()
    elif a.part == "protein":
        part_protein()
    else:
        for part in ("image", "protein"):
            subprocess.run([sys.executable, "-B", os.path.abspath(__file__), "--part", part] +
                           (["--only", a.only] if a.only else []), check=True)


if __name__ == "__main__":
    main()
