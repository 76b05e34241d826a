"""The oracle against the golden vectors captured from the reference itself.

These pin the CPU restatement (oracle/) to the reference; the GPU tests then
pin the HIP path to the oracle and to the same vectors."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import image_style as im
from oracle import protein_style as pr
from conftest import golden


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T_(a, dtype=None):
    t = torch.from_numpy(np.array(a))
    return t if dtype is None else t.to(dtype)


def build_like(g, kind, num_bits, bands=None):
    return im.build_process(T_(g["betas"]), kind, num_bits, bands, hybrid_coeff=0.001)


def check_downstream(p, g, exact=True):
    x0, t = T_(g["x0"]), T_(g["t"])
    noise, logits, noise2, w = T_(g["noise"]), T_(g["logits"]), T_(g["noise2"]), T_(g["w"])
    eq = (lambda a, b: np.testing.assert_array_equal(a.numpy(), b)) if exact else \
         (lambda a, b: np.testing.assert_allclose(a.numpy(), b, rtol=1e-6, atol=1e-7))
    eq(im.forward_probs(p, x0, t), g["q_probs"])
    x_t = im.forward_sample(p, x0, t, noise)
    np.testing.assert_array_equal(x_t.numpy(), g["x_t"])
    eq(im.posterior_logits(p, x0, x_t, t, False), g["post_true"])
    eq(im.posterior_logits(p, logits, x_t, t, True), g["post_model"])
    eq(im.model_logits(p, logits, x_t, t)[0], g["p_logits"])
    eq(im.vb_terms(p, logits, x0, x_t, t)[0], g["vb"])
    eq(im.ce_start(x0, logits), g["ce"])
    for lt in ("hybrid", "kl", "cross_entropy_x_start"):
        p.loss_type = lt
        lg = logits.clone().requires_grad_(True)
        losses = im.losses_given_xt(p, lg, x0, x_t, t)
        eq(losses.detach(), g["losses_" + lt])
        (gr,) = torch.autograd.grad((losses * w).sum(), lg)
        eq(gr, g["grad_w_" + lt])
    p.loss_type = "hybrid"
    s, probs = im.reverse_step(p, logits, x_t, t, noise2)
    np.testing.assert_array_equal(s.numpy(), g["p_sample"])
    eq(probs, g["p_sample_probs"])
    if g["prior_bpd"].size:
        eq(im.prior_term(p, x0), g["prior_bpd"])


def test_image_uniform_k32_tables_and_steps():
    g = golden("image_uniform_k32")
    p = build_like(g, "uniform", 5)
    assert sha(p.q_mats.numpy()) == str(g["q_mats_sha256"])
    assert sha(p.qT_onestep.contiguous().numpy()) == str(g["qT_onestep_sha256"])
    ts = g["slice_t"]
    np.testing.assert_array_equal(p.q_mats.numpy()[ts], g["q_mats_slices"])
    check_downstream(p, g)


def test_image_uniform_cosine_tables():
    g = golden("image_uniform_k32_cosine_tables")
    p = build_like(g, "uniform", 5)
    assert sha(p.q_mats.numpy()) == str(g["q_mats_sha256"])


def test_image_absorbing_k32():
    g = golden("image_absorbing_k32")
    p = build_like(g, "absorbing", 5)
    assert sha(p.q_mats.numpy()) == str(g["q_mats_sha256"])
    assert sha(p.qT_onestep.contiguous().numpy()) == str(g["qT_onestep_sha256"])
    check_downstream(p, g)
    # properties the reference never checked: rows sum to 1, mass ends in K//2
    np.testing.assert_allclose(p.q_mats.sum(-1).numpy(), 1.0, atol=1e-12)
    assert p.q_mats[-1, :, 16].min() > 1 - 1e-9


@pytest.mark.parametrize("name,kind,bands", [("gaussian_k16_full", "gaussian", None),
                                             ("gaussian_k16_b5", "gaussian", 5),
                                             ("uniform_k16_b3", "uniform", 3)])
def test_image_small_full_tables(name, kind, bands):
    g = golden("image_" + name)
    p = build_like(g, kind, 4, bands)
    np.testing.assert_array_equal(p.q_mats.numpy(), g["q_mats"])
    np.testing.assert_array_equal(p.qT_onestep.contiguous().numpy(), g["qT_onestep"])
    check_downstream(p, g)


def test_image_gaussian_k256_tables_and_steps():
    g = golden("image_gaussian_k256")
    p = build_like(g, "gaussian", 8)
    assert sha(p.q_mats.numpy()) == str(g["q_mats_sha256"])
    ts = g["slice_t"]
    np.testing.assert_array_equal(p.qT_onestep.contiguous().numpy()[ts], g["qT_onestep_slices"])
    check_downstream(p, g)


def test_image_misc():
    g = golden("image_misc")
    np.testing.assert_array_equal(im.make_betas(1e-4, 0.02, 1000, "linear").numpy(), g["betas_linear"])
    np.testing.assert_array_equal(im.make_betas(0, 0, 500, "cosine").numpy(), g["betas_cosine"])
    np.testing.assert_array_equal(im.make_betas(0, 0, 500, "jsd").numpy(), g["betas_jsd"])
    np.testing.assert_array_equal(im.log_min_exp(T_(g["lme_a"]), T_(g["lme_b"])).numpy(), g["lme_out"])
    with pytest.raises(NotImplementedError):
        im.make_betas(0, 1, 4, "nope")
    with pytest.raises(ValueError):
        im.build_process(torch.tensor([0.0, 0.5]), "uniform", 2)
    with pytest.raises(ValueError):
        im.build_process(torch.tensor([0.1, 0.5]), "bogus", 2)


# ---------------------------------------------------------------------------
def test_protein_tables():
    g = golden("protein_tables_k22")
    for s in ("sohl", "linear", "cosine"):
        name = {"sohl": "sohl-dickstein"}.get(s, s)
        np.testing.assert_array_equal(pr.beta_schedule(500, schedule=name).numpy(), g["betas_" + s])
    qb, qt = pr.uniform_tables(22, 500, "sohl-dickstein")
    assert sha(qb.numpy()) == str(g["Q_bar_sha256"])
    assert sha(qt.numpy()) == str(g["Q_t_sha256"])
    g2 = golden("protein_tables_k22_linear60")
    qb, qt = pr.uniform_tables(22, 60, "linear")
    np.testing.assert_array_equal(qb.numpy(), g2["Q_bar"])
    np.testing.assert_array_equal(qt.numpy(), g2["Q_t"])
    # the float32 off-diagonal quirk: rows do not sum to exactly one
    assert abs(qt[0].sum(-1) - 1).max() > 1e-9


def test_protein_forward_and_lvb():
    g = golden("protein_lvb_blat")
    qb, qt = pr.uniform_tables(22, 500, "sohl-dickstein")
    x0, x_t, ts = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"])
    for i in range(x0.shape[0]):
        s, p = pr.corrupt(pr.one_hot(x0[i]), qb[min(int(ts[i]), 499)], T_(g["expo"][i]))
        np.testing.assert_array_equal(s.numpy(), g["x_t"][i])
        np.testing.assert_array_equal(p.numpy(), g["q"][i])
    for n_tok in (22, 24):
        pred = T_(g[f"pred{n_tok}"]).requires_grad_(True)
        args = (pr.one_hot(x_t), T_(g["q"]), pred, x0, pr.one_hot(x0), ts, qt, qb)
        per = pr.lvb_per_sequence(*args)
        np.testing.assert_array_equal(per.detach().numpy(), g[f"per_seq{n_tok}"])
        total = pr.lvb(*args)
        np.testing.assert_array_equal(total.detach().numpy(), g[f"lvb{n_tok}"])
        (gr,) = torch.autograd.grad(total, pred)
        np.testing.assert_array_equal(gr.numpy(), g[f"grad{n_tok}"])


def test_protein_lvb_msa():
    g = golden("protein_lvb_msa")
    qb, qt = pr.uniform_tables(22, 500, "sohl-dickstein")
    pred = T_(g["pred"]).requires_grad_(True)
    x0, x_t = T_(g["x0"]), T_(g["x_t"])
    total = pr.lvb(pr.one_hot(x_t), T_(g["q"]), pred, x0, pr.one_hot(x0), T_(g["timestep"]), qt, qb, msa=True)
    np.testing.assert_array_equal(total.detach().numpy(), g["lvb"])
    (gr,) = torch.autograd.grad(total, pred)
    np.testing.assert_array_equal(gr.numpy(), g["grad"])


def test_protein_driver_loss():
    """(lvb + lambda * ce) * n_tokens of the training loop (notebook cell-12 / cell-13): value, gradient and the
    cross-entropy term itself, single sequences (22 and 24 logits) and MSAs."""
    g, h = golden("protein_lvb_blat"), golden("protein_driver_loss")
    qb, qt = pr.uniform_tables(22, 500, "sohl-dickstein")
    x0, x_t, ts = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"])
    for n_tok in (22, 24):
        for lam in (0.001, 0.5):
            pred = T_(g[f"pred{n_tok}"]).requires_grad_(True)
            loss, _, ce = pr.driver_loss(pr.one_hot(x_t), T_(g["q"]), pred, x0, pr.one_hot(x0), ts, qt, qb, lam)
            np.testing.assert_array_equal(loss.detach().numpy(), h[f"loss{n_tok}_lam{lam}"])
            np.testing.assert_array_equal(ce.detach().numpy(), g[f"ce{n_tok}"])
            (gr,) = torch.autograd.grad(loss, pred)
            np.testing.assert_array_equal(gr.numpy(), h[f"grad{n_tok}_lam{lam}"])
    m = golden("protein_lvb_msa")
    x0, x_t = T_(m["x0"]), T_(m["x_t"])
    for lam in (0.001, 0.5):
        pred = T_(m["pred"]).requires_grad_(True)
        loss, _, ce = pr.driver_loss(pr.one_hot(x_t), T_(m["q"]), pred, x0, pr.one_hot(x0), T_(m["timestep"]), qt, qb,
                                     lam, msa=True)
        np.testing.assert_array_equal(loss.detach().numpy(), h[f"msa_loss_lam{lam}"])
        np.testing.assert_array_equal(ce.detach().numpy(), h["msa_ce"])
        (gr,) = torch.autograd.grad(loss, pred)
        np.testing.assert_array_equal(gr.numpy(), h[f"msa_grad_lam{lam}"])


def test_protein_generate_replay():
    g = golden("protein_generate")
    emb, tvec = T_(g["emb"]), T_(g["tvec"])
    model = lambda x, tt: emb[x] + tvec[tt][:, None, :]
    expo = T_(g["expo"])
    sample, strings = pr.generate(model, T_(g["Q_t"]), T_(g["Q_bar"]), 20, T_(g["x_init"]),
                                  lambda step, i: expo[step, i])
    np.testing.assert_array_equal(sample.numpy(), g["sample"])
    assert strings == [str(s) for s in g["strings"]]
    # one batched step equals the per-sequence loop
    x = T_(g["x_init"])
    one = pr.reverse_step(model(x, torch.tensor([19] * 3)), x, 19, T_(g["Q_t"]), T_(g["Q_bar"]), expo[0])
    assert one.shape == x.shape


def test_protein_generate_msa_replay():
    g = golden("protein_generate_msa")
    emb, tvec = T_(g["emb"]), T_(g["tvec"])
    model = lambda x, tt: emb[x] + tvec[tt][:, None, None, :]
    ea, eq = T_(g["expo_all"]), T_(g["expo_query"])
    sample, strings = pr.generate_msa(model, T_(g["Q_t"]), T_(g["Q_bar"]), 12, T_(g["x_init"]),
                                      lambda step: (ea[step], eq[step]), penalty_value=float(g["penalty"]))
    np.testing.assert_array_equal(sample.numpy(), g["sample"])
    assert [s[0] for s in strings] == [str(s) for s in g["strings"]]
    # the gap class never appears in a query row
    assert int((sample[:, 0] == 21).sum()) == 0


def test_protein_collate_v2():
    g = golden("protein_collate_v2")
    tables = golden("protein_tables_k22")
    Qb, _ = pr.uniform_tables(22, 500, "sohl-dickstein")
    np.testing.assert_array_equal(Qb.numpy()[tables["slice_t"]], tables["Q_bar_slices"])
    toks = T_(g["tokens"])
    for t in (1, 37, 500):
        src, ts, qx = pr.collate_v2(toks, t, 500, Qb, T_(g[f"expo_t{t}"]))
        np.testing.assert_array_equal(src.numpy(), g[f"src_t{t}"])
        np.testing.assert_array_equal(ts.numpy(), g[f"timesteps_t{t}"])
        np.testing.assert_array_equal(qx.numpy(), g[f"q_t{t}"])
    assert g["timesteps_t500"].tolist() == [500, 499, 499, 499]      # the reference's decrement after the first


def test_protein_collaters():
    """Collater / CollaterMSA of the reference's dataset.py (executed as it is by tools/make_golden.py): timestep
    draw order, corrupted tokens, probabilities, the one-hot tensors and the doubled q_x of the MSA collater."""
    g = golden("protein_collaters")
    qb, _ = pr.uniform_tables(22, 500, "sohl-dickstein")
    toks = T_(g["tokens"])[:8]
    np.random.seed(int(g["np_seed"]))
    src, src_oh, ts, tgt, tgt_oh, qx = pr.collate(toks, 500, qb, T_(g["expo"]))
    np.testing.assert_array_equal(ts.numpy(), g["timesteps"])
    np.testing.assert_array_equal(src.numpy(), g["src"])
    np.testing.assert_array_equal(tgt.numpy(), g["tokenized"])
    np.testing.assert_array_equal(qx.numpy()[:, :4], g["q_rows"])
    assert sha(src_oh.numpy()) == str(g["src_one_hot_sha256"]) and sha(tgt_oh.numpy()) == str(g["tgt_one_hot_sha256"])
    msa = T_(g["tokens"])[:12, :60].reshape(3, 4, 60)
    np.random.seed(int(g["msa_np_seed"]))
    src, src_oh, ts, tgt, tgt_oh, qx = pr.collate_msa(msa, 500, qb, T_(g["msa_expo"]))
    np.testing.assert_array_equal(ts.numpy(), g["msa_timesteps"])
    np.testing.assert_array_equal(src.numpy(), g["msa_src"])
    np.testing.assert_array_equal(tgt.numpy(), g["msa_tokenized"])
    assert tuple(qx.shape) == tuple(g["msa_q_shape"]) == (6, 4, 60, 22)
    np.testing.assert_array_equal(qx.numpy()[:, :, :3], g["msa_q_rows"])
    assert sha(src_oh.numpy()) == str(g["msa_src_one_hot_sha256"])


def _mutant_fixture():
    g = golden("protein_mutant_scores")
    rows = g["expo_rows"].tolist()
    L = len([c for c in str(g["focus_seq"]) if c == c.upper()])
    flat, Em, o = g["expo"], [], 0
    for r in rows:
        Em.append(torch.from_numpy(flat[o:o + r * L * 22].reshape(r, L, 22)))
        o += r * L * 22
    return g, Em


def test_protein_mutant_scores(tmp_path):
    g, Em = _mutant_fixture()
    focus = str(g["focus_seq"])
    start = int(str(g["focus_name"]).split("/")[-1].split("-")[0])
    # sequences as the reference assembles them (utils.py:107-150), restated for the fixture's CSV
    trimmed = [c for c in focus if c == c.upper()]
    col = {start + i: k for k, i in enumerate(i for i, c in enumerate(focus) if c == c.upper())}
    seqs, desc = ["".join(trimmed)], []
    for line in str(g["csv"]).splitlines()[1:]:
        muts = line.split(",")[0].split(":")
        if all(int(m[1:-1]) in col and focus[int(m[1:-1]) - start] == m[0] and m[0] != m[-1] for m in muts):
            s = list(trimmed)
            for m in muts:
                s[col[int(m[1:-1])]] = m[-1]
            seqs.append("".join(s)); desc.append(":".join(muts))
    assert desc == [str(d) for d in g["descriptors"]]
    toks = torch.from_numpy(np.stack([pr.tokenize(s) for s in seqs]))
    emb, tvec = T_(g["emb"]), T_(g["tvec"])
    model = lambda x, tt: emb[x] + tvec[tt][:, None, :]
    n_iter, mb, steps = int(g["n_iter"]), int(g["minibatch"]), int(g["steps"])
    nb = (len(seqs) + mb - 1) // mb
    tmid = g["t_mid"].tolist()
    delta = pr.mutant_scores(toks, model, T_(g["Q_t"]), T_(g["Q_bar"]), steps, n_iter, mb,
                             lambda i: g["orders"][i], lambda i, j: tmid[i * nb + j // mb],
                             lambda i, j, h: Em[(i * nb + j // mb) * 3 + h])
    np.testing.assert_allclose(delta, g["delta_elbos"], rtol=1e-6, atol=1e-7)


def test_logistic_head_and_bpd_loop():
    """The discretised-logistic head (bit for bit, and its autograd gradient) and
    calc_bpd_loop replayed on the recorded per-step noise."""
    g = golden("image_misc")
    loc, ls = T_(g["logistic_loc"]).requires_grad_(True), T_(g["logistic_log_scale"]).requires_grad_(True)
    out = im.logits_from_logistic_pars(loc, ls, 16)
    np.testing.assert_array_equal(out.detach().numpy(), g["logistic_logits"])
    (out * T_(g["logistic_w"])).sum().backward()
    np.testing.assert_array_equal(loc.grad.numpy(), g["logistic_grad_loc"])
    np.testing.assert_array_equal(ls.grad.numpy(), g["logistic_grad_log_scale"])
    g = golden("image_bpd_loop")
    p = im.build_process(T_(g["betas"]), "gaussian", 4, None, hybrid_coeff=0.001)
    logits = im.logits_from_logistic_pars(T_(g["loc"]), T_(g["log_scale"]), 16)
    res = im.bpd_sweep(p, lambda x, t: logits, T_(g["x0"]), T_(g["noises"]))
    for k in ("total", "vbterms", "prior"):
        np.testing.assert_array_equal(res[k].numpy(), g[k])


@pytest.mark.parametrize("name,kind,bits", [("image_uniform_k32", "uniform", 5), ("image_absorbing_k32", "absorbing", 5),
                                            ("image_gaussian_k256", "gaussian", 8)])
def test_float32_rounding_floor_of_the_reference_gradient(name, kind, bits):
    """How far the reference's own float32 gradient (golden vectors) is from a float64 evaluation of the same
    formulas on the same float32 table values: a few 1e-7 of the largest element, absolutely, whatever the
    element's size.  The GPU tests hold gradients to 1e-5 relative above GRAD_FLOOR = 1e-6 of the largest element;
    this test shows that floor to be within a small factor of what float32 leaves undefined, not a loosened bar.
    Per-example losses, by contrast, are defined to 1e-7 relative."""
    g = golden(name)
    x0, t, xt, w = T_(g["x0"]), T_(g["t"]), T_(g["x_t"]), T_(g["w"])
    worst = 0.0
    for lt in ("hybrid", "kl"):
        p = im.build_process(T_(g["betas"]), kind, bits, None, hybrid_coeff=0.001, loss_type=lt)
        lg = T_(g["logits"]).double().requires_grad_(True)
        with im.arithmetic(torch.float64):
            losses = im.losses_given_xt(p, lg, x0, xt, t)
            (losses * w.double()).sum().backward()
        exact, ref = lg.grad.numpy(), g["grad_w_" + lt].astype(np.float64)
        worst = max(worst, np.abs(ref - exact).max() / np.abs(exact).max())
        np.testing.assert_allclose(g["losses_" + lt], losses.detach().numpy(), rtol=2e-7, atol=5e-7)
    assert 1e-7 < worst < 5e-7, worst
