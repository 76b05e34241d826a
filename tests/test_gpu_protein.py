"""GPU parity of the protein-style process against the golden vectors captured
from the reference (tables, forward corruption, LVB loss + gradient in both
variants, full reverse-sampling replay) and against the CPU oracle.

Bars: tokens bit-exact with the recorded Exp(1) draws; float64 losses and
float32 gradients within 1e-5 relative."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import protein_style as pr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# Gradients are float32 tensors (the reference casts its float64 gradient to the float32 logits): an element is
# p_j (g_j - sum_i p_i g_i), a difference of terms of the size of the largest element, which the training kernel
# forms in float32.  Elements are held to 1e-5 relative above this fraction of the largest element.
GRAD_FLOOR = 1e-6


@pytest.fixture(scope="module")
def pkg():
    import d3pm_amd
    return d3pm_amd


@pytest.fixture(scope="module")
def tables500(pkg):
    tok = pkg.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=500, schedule="sohl-dickstein", device=DEV)
    return tok, Qb, Qt


def T_(a, dev=DEV):
    return torch.from_numpy(np.array(a)).to(dev)


def test_tables_golden(pkg, tables500):
    tok, Qb, Qt = tables500
    g = golden("protein_tables_k22")
    ts = g["slice_t"]
    assert Qb.dtype == torch.float64 and Qb.shape == (500, 22, 22)
    np.testing.assert_array_equal(Qt.cpu().numpy()[ts], g["Q_t_slices"])      # float32 scalars promoted: exact
    np.testing.assert_allclose(Qb.cpu().numpy()[ts], g["Q_bar_slices"], rtol=1e-12, atol=1e-15)
    g2 = golden("protein_tables_k22_linear60")
    Qb2, Qt2 = tok.q_uniform_schedule(num_steps=60, schedule="linear", device=DEV)
    np.testing.assert_array_equal(Qt2.cpu().numpy(), g2["Q_t"])
    np.testing.assert_allclose(Qb2.cpu().numpy(), g2["Q_bar"], rtol=1e-12, atol=1e-15)
    # cumprod_matrix on a list of matrices, as the reference calls it
    bars = pkg.cumprod_matrix([m for m in T_(g2["Q_t"])])
    assert len(bars) == 60
    np.testing.assert_allclose(torch.stack(bars).cpu().numpy(), g2["Q_bar"], rtol=1e-12, atol=1e-15)


def test_forward_corruption_golden(pkg, tables500):
    from d3pm_amd.protein_generation.utils import corrupt_batch, sample_transition_matrix
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    g = golden("protein_lvb_blat")
    x0, ts = T_(g["x0"]), T_(g["timestep"])
    tb = protein_tables(Qt, Qb)
    x_t, q = corrupt_batch(tb, x0, ts.clamp(max=499), T_(g["expo"]), want_probs=True)
    np.testing.assert_array_equal(x_t.cpu().numpy(), g["x_t"])
    np.testing.assert_allclose(q.cpu().numpy(), g["q"], rtol=1e-12, atol=1e-18)
    # one sequence through the reference-shaped call
    s, p = sample_transition_matrix(tok.one_hot(x0[2]), Qb[250], noise=T_(g["expo"][2]))
    np.testing.assert_array_equal(s.cpu().numpy(), g["x_t"][2])
    assert p.shape == (263, 22)
    # Philox path: valid tokens, repeatable under a seed, right marginal at t = 499
    torch.manual_seed(5); a, _ = corrupt_batch(tb, x0, ts.clamp(max=499))
    torch.manual_seed(5); b, _ = corrupt_batch(tb, x0, ts.clamp(max=499))
    assert torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < 22
    big = torch.zeros((64, 4096), dtype=torch.int64, device=DEV)
    xt, _ = corrupt_batch(tb, big, torch.full((64,), 499, device=DEV))
    freq = torch.bincount(xt.flatten(), minlength=22).double().cpu().numpy() / xt.numel()
    P = Qb[499, 0].cpu().numpy()                       # nearly, not exactly, uniform: the start class keeps 0.0474
    assert (np.abs(freq - P) < 5 * np.sqrt(P * (1 - P) / xt.numel())).all(), (freq, P)


@pytest.mark.parametrize("want_probs", [False, True])
def test_corrupt_device_noise_distribution(pkg, tables500, want_probs):
    """Forward corruption with noise drawn on the device (token-per-lane kernel without probabilities, lane-group
    kernel with them): class frequencies of x_t against row x0 of Qbar_t, at timesteps from 'hardly anything moves'
    to the uniform end, a different timestep and start class per sequence."""
    from scipy.stats import binom
    from d3pm_amd.protein_generation.utils import corrupt_batch
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    ts = [0, 1, 5, 40, 150, 499]
    Bn, Ln = len(ts) * 4, 50000
    cls = torch.tensor([(3 * i + 1) % 22 for i in range(Bn)])
    x0 = cls[:, None].expand(Bn, Ln).contiguous().to(DEV)
    t = torch.tensor(ts).repeat_interleave(4).to(DEV)
    torch.manual_seed(3)
    xt, probs = corrupt_batch(tb, x0, t, want_probs=want_probs)
    torch.manual_seed(3)
    again, _ = corrupt_batch(tb, x0, t, want_probs=want_probs)
    assert torch.equal(xt, again)
    assert int(xt.min()) >= 0 and int(xt.max()) < 22
    counts = torch.nn.functional.one_hot(xt.cpu(), 22).sum(dim=1).numpy()          # [B, K]
    P = Qb.cpu()[t.cpu(), cls].numpy()                                               # [B, K]
    if want_probs:
        np.testing.assert_allclose(probs[:, 0].cpu().numpy(), P, rtol=1e-12, atol=1e-18)
    tail = np.minimum(binom.cdf(counts, Ln, P), binom.sf(counts - 1, Ln, P))
    assert tail.min() > 1e-7, (tail.min(), np.unravel_index(tail.argmin(), tail.shape))
    # pooled over the four sequences of a timestep: the share of moved tokens
    for i, tt in enumerate(ts):
        rows = slice(4 * i, 4 * i + 4)
        moved = float((xt[rows] != x0[rows]).double().mean())
        expect = float(np.mean([1 - P[r, cls[r]] for r in range(4 * i, 4 * i + 4)]))
        assert abs(moved - expect) < 6 * np.sqrt(expect * (1 - expect) / (4 * Ln)) + 1e-7, (tt, moved, expect)


@pytest.mark.parametrize("n_tok", [22, 24])
def test_lvb_loss_golden(pkg, tables500, n_tok):
    tok, Qb, Qt = tables500
    g = golden("protein_lvb_blat")
    x0, x_t, ts, q = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"]), T_(g["q"])
    pred = T_(g[f"pred{n_tok}"]).requires_grad_(True)
    per = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)(
        tok.one_hot(x_t), q, pred, x0, tok.one_hot(x0), ts, Qt, Qb)
    assert per.shape == (8,) and per.dtype == torch.float64
    np.testing.assert_allclose(per.detach().cpu().numpy(), g[f"per_seq{n_tok}"], rtol=1e-5, atol=1e-9)
    lvb = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok)(tok.one_hot(x_t), q, pred, x0, tok.one_hot(x0), ts, Qt, Qb)
    np.testing.assert_allclose(lvb.detach().cpu().numpy(), g[f"lvb{n_tok}"], rtol=1e-5)
    (gr,) = torch.autograd.grad(lvb, pred)
    ref = g[f"grad{n_tok}"]
    np.testing.assert_allclose(gr.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())
    # the notebook's scaling: (lvb + lambda*ce) * n_tokens flows back through the baked 1/B
    pred2 = T_(g[f"pred{n_tok}"]).requires_grad_(True)
    lvb2 = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok)(None, q, pred2, x0, None, ts, Qt, Qb, src=x_t)
    (lvb2 * 7.5).backward()
    np.testing.assert_allclose(pred2.grad.cpu().numpy(), 7.5 * ref, rtol=1e-5, atol=7.5 * GRAD_FLOOR * np.abs(ref).max())


def test_lvb_backward_twice_and_msa_collater_input(pkg, tables500):
    """(1) A second backward through a retained graph recomputes the gradient (the reference supports it).
    (2) CollaterMSA's default output -- q_x appended twice, [2B, D, L, K] -- goes straight into D3PMLVBLossMSA, as
    in the reference, which only reads q[i] for i < B; a t == tmax row makes the loss read it."""
    from d3pm_amd.protein_generation.dataset import CollaterMSA
    tok, Qb, Qt = tables500
    g = golden("protein_lvb_blat")
    x0, x_t, ts, q = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"]), T_(g["q"])
    pred = T_(g["pred24"]).requires_grad_(True)
    per = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)(tok.one_hot(x_t), q, pred, x0, tok.one_hot(x0), ts, Qt, Qb)
    w1, w2 = torch.rand(8, device=DEV, dtype=torch.float64), torch.rand(8, device=DEV, dtype=torch.float64)
    g1, = torch.autograd.grad((per * w1).sum(), pred, retain_graph=True)
    g1 = g1.clone()
    g2, = torch.autograd.grad((per * w2).sum(), pred, retain_graph=True)
    ones, = torch.autograd.grad(per.sum(), pred)
    for got, w in ((g1, w1), (g2, w2)):
        want = (ones * w.float()[:, None, None]).cpu().numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-7 * np.abs(want).max())
    # collater -> MSA loss
    msas = [["ACDEFGHIKL", "ACDEFGHIKV", "MCDEFGH-KL"], ["LKIHGFEDCA", "LKIHGFEDCM", "LKIHG-EDCA"]]
    coll = CollaterMSA(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb, device=DEV)
    src, src_oh, tsm, tgt, tgt_oh, Qc, Qbc, q_x = coll(msas)
    assert q_x.shape[0] == 2 * len(msas)
    tsm = tsm.clone()
    tsm[0] = 500                                              # the branch that reads q
    pm = torch.randn(tuple(tgt.shape) + (24,), device=DEV, requires_grad=True)
    loss = pkg.D3PMLVBLossMSA(tmax=500, tokenizer=tok)(src_oh, q_x, pm, tgt, tgt_oh, tsm, Qc, Qbc)
    want = pkg.D3PMLVBLossMSA(tmax=500, tokenizer=tok)(src_oh, q_x[:len(msas)], pm, tgt, tgt_oh, tsm, Qc, Qbc)
    assert torch.isfinite(loss) and torch.equal(loss, want)
    loss.backward()
    assert torch.isfinite(pm.grad).all()


def test_lvb_loss_msa_golden(pkg, tables500):
    tok, Qb, Qt = tables500
    g = golden("protein_lvb_msa")
    x0, x_t, ts, q = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"]), T_(g["q"])
    pred = T_(g["pred"]).requires_grad_(True)
    lvb = pkg.D3PMLVBLossMSA(tmax=500, tokenizer=tok)(tok.one_hot(x_t), q, pred, x0, tok.one_hot(x0), ts, Qt, Qb)
    np.testing.assert_allclose(lvb.detach().cpu().numpy(), g["lvb"], rtol=1e-5)
    (gr,) = torch.autograd.grad(lvb, pred)
    np.testing.assert_allclose(gr.cpu().numpy(), g["grad"], rtol=1e-5, atol=GRAD_FLOOR * np.abs(g["grad"]).max())


@pytest.mark.parametrize("n_tok,B,D,L", [(22, 8, 3, 50), (24, 8, 4, 64), (24, 200, 2, 37)])
def test_lvb_loss_msa_against_oracle(pkg, tables500, n_tok, B, D, L):
    """D3PMLVBLossMSA's training form (float32 kernel: rho (log rho + KL) per token) against the CPU oracle:
    per-sequence losses and the gradient, early and late timesteps, the t == 1 and t == tmax branches, position
    counts that are no multiple of a wave, a chunked launch (B = 8) and one workgroup per element (B = 200)."""
    from d3pm_amd._ops import ops
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    gen = torch.Generator().manual_seed(5 + n_tok + B)
    x0 = torch.randint(0, 22, (B, D, L), generator=gen)
    ts = torch.randint(2, 500, (B,), generator=gen)
    ts[:8] = torch.tensor([1, 2, 3, 17, 250, 499, 500, 2])
    Qb_c, Qt_c = Qb.cpu(), Qt.cpu()
    # x_t drawn from the forward process, so that every (x0, x_t) pair has weight under Qbar_{t-1} too
    rows = Qb_c[(ts - 1).clamp(max=499)][:, None, None].expand(B, D, L, 22, 22)
    probs = torch.gather(rows, 3, x0[..., None, None].expand(B, D, L, 1, 22))[..., 0, :]
    x_t = torch.multinomial(probs.reshape(-1, 22), 1, generator=gen).reshape(B, D, L)
    q = torch.softmax(torch.randn(B, D, L, 22, generator=gen, dtype=torch.float64), -1)
    pred = torch.randn(B, D, L, n_tok, generator=gen) * 2.0
    pr_ref = pred.clone().requires_grad_(True)
    per_ref = pr.lvb_per_sequence(pr.one_hot(x_t, 22), q, pr_ref, x0, pr.one_hot(x0, 22), ts, Qt_c, Qb_c,
                                  K=22, tmax=500, msa=True)
    baked = float(np.float32(1.0) / np.float32(B))
    (per_ref.sum() * baked).backward()
    tb = protein_tables(Qt, Qb)
    per, dp = ops.lvb_loss(*tb.op_args(), pred.to(DEV).reshape(B, D * L, n_tok).contiguous(), x0.to(DEV), x_t.to(DEV),
                           ts.to(DEV), q.to(DEV), 500, D, baked, True)
    np.testing.assert_allclose(per.cpu().numpy(), per_ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    ref = pr_ref.grad.numpy().reshape(B, D * L, n_tok)
    np.testing.assert_allclose(dp.cpu().numpy().reshape(ref.shape), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())
    # and through the reference-shaped module
    pm = pred.to(DEV).requires_grad_(True)
    lvb = pkg.D3PMLVBLossMSA(tmax=500, tokenizer=tok)(None, q.to(DEV), pm, x0.to(DEV), None, ts.to(DEV), Qt, Qb,
                                                      src=x_t.to(DEV))
    np.testing.assert_allclose(float(lvb.detach()), float(per_ref.detach().sum() / B), rtol=1e-5)


@pytest.mark.parametrize("n_tok", [22, 24])
@pytest.mark.parametrize("lam", [0.001, 0.5])
def test_driver_loss_golden(pkg, tables500, n_tok, lam):
    """The training loop's loss, (lvb + lambda * ce) * n_tokens (notebook cell-12), from the module with
    hybrid_coeff / out_scale: value, cross-entropy term and gradient against the reference's, one launch."""
    tok, Qb, Qt = tables500
    g, h = golden("protein_lvb_blat"), golden("protein_driver_loss")
    x0, x_t, ts, q = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"]), T_(g["q"])
    pred = T_(g[f"pred{n_tok}"]).requires_grad_(True)
    n_tokens = x0.shape[0] * x0.shape[1]
    lf = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, hybrid_coeff=lam)
    loss = lf(None, q, pred, x0, None, ts, Qt, Qb, src=x_t, out_scale=n_tokens)
    np.testing.assert_allclose(float(loss.detach()), h[f"loss{n_tok}_lam{lam}"], rtol=1e-5)
    np.testing.assert_allclose(float(lf.ce_loss), g[f"ce{n_tok}"], rtol=1e-5)
    loss.backward()
    ref = h[f"grad{n_tok}_lam{lam}"]
    np.testing.assert_allclose(pred.grad.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())
    # the notebook's own spelling: the module's value times n_tokens outside (upstream gradient != baked scale)
    pred2 = T_(g[f"pred{n_tok}"]).requires_grad_(True)
    lf2 = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, hybrid_coeff=lam)
    (lf2(tok.one_hot(x_t), q, pred2, x0, tok.one_hot(x0), ts, Qt, Qb) * n_tokens).backward()
    np.testing.assert_allclose(pred2.grad.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())
    # forward-only evaluation (float64 kernel) carries the term too: per-sequence lvb + lambda * ce
    with torch.no_grad():
        lf3 = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True, hybrid_coeff=lam)
        per = lf3(None, q, pred, x0, None, ts, Qt, Qb, src=x_t)
    want = g[f"per_seq{n_tok}"] + lam * lf3.ce_seq.cpu().numpy()
    np.testing.assert_allclose(per.cpu().numpy(), want, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(float(lf3.ce_loss), g[f"ce{n_tok}"], rtol=1e-6)


@pytest.mark.parametrize("lam", [0.001, 0.5])
def test_driver_loss_msa_golden(pkg, tables500, lam):
    """Notebook cell-13: the same loop around D3PMLVBLossMSA, n_tokens = B * D."""
    tok, Qb, Qt = tables500
    g, h = golden("protein_lvb_msa"), golden("protein_driver_loss")
    x0, x_t, ts, q = T_(g["x0"]), T_(g["x_t"]), T_(g["timestep"]), T_(g["q"])
    pred = T_(g["pred"]).requires_grad_(True)
    n_tokens = x0.shape[0] * x0.shape[1]
    lf = pkg.D3PMLVBLossMSA(tmax=500, tokenizer=tok, hybrid_coeff=lam)
    loss = lf(tok.one_hot(x_t), q, pred, x0, tok.one_hot(x0), ts, Qt, Qb, out_scale=n_tokens)
    np.testing.assert_allclose(float(loss.detach()), h[f"msa_loss_lam{lam}"], rtol=1e-5)
    np.testing.assert_allclose(float(lf.ce_loss), h["msa_ce"], rtol=1e-5)
    loss.backward()
    ref = h[f"msa_grad_lam{lam}"]
    np.testing.assert_allclose(pred.grad.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())


@pytest.mark.parametrize("B,L,n_tok,msa", [(1030, 270, 24, 0), (400, 263, 22, 0), (256, 512, 24, 0), (200, 37 * 2, 24, 2),
                                           (9, 300, 23, 0), (210, 41 * 3, 22, 3)])
def test_driver_loss_against_oracle(pkg, tables500, B, L, n_tok, msa):
    """d3pm_lvb_hybrid_loss at the launch shapes of the training kernel (one workgroup per element, 256-thread
    workgroups walking several passes, chunks summed by a second launch, the MSA variant) and on the float64 kernel
    (23 logits): per-sequence LVB, per-sequence cross-entropy and the gradient of
    grad_scale * sum per_seq + ce_grad_scale * sum ce_seq against the CPU oracle, every timestep branch present.
    Logits beyond the K classes are made large in some rows: the cross-entropy runs over all of them."""
    from d3pm_amd._ops import ops
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    gen = torch.Generator().manual_seed(B + n_tok)
    x0 = torch.randint(0, 22, (B, L), generator=gen)
    ts = torch.randint(1, 501, (B,), generator=gen)
    ts[:4] = torch.tensor([1, 2, 500, 499])
    Qb_c, Qt_c = Qb.cpu(), Qt.cpu()
    probs = Qb_c[(ts - 1).clamp(max=499)][torch.arange(B)[:, None], x0]                   # [B, L, K]
    x_t = torch.multinomial(probs.reshape(-1, 22), 1, generator=gen).reshape(B, L)
    q = torch.softmax(torch.randn(B, L, 22, generator=gen, dtype=torch.float64), -1)
    pred = torch.randn(B, L, n_tok, generator=gen) * 2.0
    if n_tok > 22:
        pred[::3, ::5, 22:] += 9.0                     # further logits above every class logit
        pred[1::3, 1::5, 22:] -= 30.0
    shape = (B, msa, L // msa) if msa else (B, L)
    rs = lambda a: a.reshape(shape + a.shape[2:])
    pr_ref = pred.clone().requires_grad_(True)
    per_ref = pr.lvb_per_sequence(pr.one_hot(rs(x_t), 22), rs(q), rs(pr_ref), rs(x0), pr.one_hot(rs(x0), 22), ts,
                                  Qt_c, Qb_c, K=22, tmax=500, msa=bool(msa))
    ce_tok = torch.nn.functional.cross_entropy(pr_ref.reshape(-1, n_tok), x0.reshape(-1), reduction="none")
    ce_ref = ce_tok.reshape(B, L).double().mean(dim=1)
    gs, cgs = float(np.float32(3.0) / np.float32(B)), float(np.float32(0.7) / np.float32(B))
    (per_ref.sum() * gs + ce_ref.sum() * cgs).backward()
    tb = protein_tables(Qt, Qb)
    per, ce, dp = ops.lvb_hybrid_loss(*tb.op_args(), pred.to(DEV), x0.to(DEV), x_t.to(DEV), ts.to(DEV), q.to(DEV), 500,
                                      msa, gs, cgs, True)
    np.testing.assert_allclose(per.cpu().numpy(), per_ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ce.cpu().numpy(), ce_ref.detach().numpy(), rtol=1e-5)
    ref = pr_ref.grad.numpy()
    np.testing.assert_allclose(dp.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())
    # forward only (float64 kernel): the same two sums
    per2, ce2, _ = ops.lvb_hybrid_loss(*tb.op_args(), pred.to(DEV), x0.to(DEV), x_t.to(DEV), ts.to(DEV), q.to(DEV),
                                       500, msa, gs, cgs, False)
    np.testing.assert_allclose(per2.cpu().numpy(), per_ref.detach().numpy(), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(ce2.cpu().numpy(), ce_ref.detach().numpy(), rtol=1e-6)


@pytest.mark.parametrize("B,L", [(1030, 270), (400, 263), (257, 513)])
def test_lvb_loss_large_batch_against_oracle(pkg, tables500, B, L):
    """Batches beyond one workgroup per CU run the training kernel in 256-thread workgroups whose waves walk
    several passes (B = 1030: one chunk of two passes, the second ragged) or in chunks summed by a second launch
    (B = 400): per-sequence losses and the gradient against the CPU oracle."""
    from d3pm_amd._ops import ops
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    gen = torch.Generator().manual_seed(B)
    x0 = torch.randint(0, 22, (B, L), generator=gen)
    ts = torch.randint(1, 501, (B,), generator=gen)
    ts[:4] = torch.tensor([1, 2, 500, 499])
    Qb_c, Qt_c = Qb.cpu(), Qt.cpu()
    probs = Qb_c[(ts - 1).clamp(max=499)][torch.arange(B)[:, None], x0]                   # [B, L, K]
    x_t = torch.multinomial(probs.reshape(-1, 22), 1, generator=gen).reshape(B, L)
    q = torch.softmax(torch.randn(B, L, 22, generator=gen, dtype=torch.float64), -1)
    pred = torch.randn(B, L, 24, generator=gen) * 2.0
    pr_ref = pred.clone().requires_grad_(True)
    per_ref = pr.lvb_per_sequence(pr.one_hot(x_t, 22), q, pr_ref, x0, pr.one_hot(x0, 22), ts, Qt_c, Qb_c,
                                  K=22, tmax=500)
    baked = float(np.float32(1.0) / np.float32(B))
    (per_ref.sum() * baked).backward()
    tb = protein_tables(Qt, Qb)
    per, dp = ops.lvb_loss(*tb.op_args(), pred.to(DEV), x0.to(DEV), x_t.to(DEV), ts.to(DEV), q.to(DEV), 500, 0, baked,
                           True)
    np.testing.assert_allclose(per.cpu().numpy(), per_ref.detach().numpy(), rtol=1e-5, atol=1e-9)
    ref = pr_ref.grad.numpy()
    np.testing.assert_allclose(dp.cpu().numpy(), ref, rtol=1e-5, atol=GRAD_FLOOR * np.abs(ref).max())


@pytest.mark.parametrize("want_grad", [False, True])
def test_lvb_loss_without_q(pkg, tables500, want_grad):
    """q = None (the collaters' lean form): an element at t == tmax has no probabilities to read and gets NaN, every
    other element its loss -- no read through the null pointer (float64 evaluation kernel and float32 training
    kernel)."""
    tok, Qb, Qt = tables500
    gen = torch.Generator().manual_seed(8)
    x0 = torch.randint(0, 22, (6, 70), generator=gen).to(DEV)
    x_t = torch.randint(0, 22, (6, 70), generator=gen).to(DEV)
    ts = torch.tensor([500, 1, 2, 250, 499, 500], device=DEV)
    pred = torch.randn(6, 70, 24, generator=gen).to(DEV).requires_grad_(want_grad)
    per = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)(None, None, pred, x0, None, ts, Qt, Qb, src=x_t)
    per = per.detach().cpu().numpy()
    assert np.isnan(per[[0, 5]]).all() and np.isfinite(per[1:5]).all(), per
    q = torch.softmax(torch.randn(6, 70, 22, generator=gen, dtype=torch.float64), -1).to(DEV)
    full = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)(None, q, pred, x0, None, ts, Qt, Qb, src=x_t)
    np.testing.assert_allclose(per[1:5], full.detach().cpu().numpy()[1:5], rtol=1e-12 if not want_grad else 1e-5)


def test_protein_tables_hit_on_fresh_uploads(pkg):
    """The reference's loop uploads Q and Q_bar anew on every step (notebook cell-12: Q.to(device)) and hands the
    copies to the loss: fresh pointers, known content.  The digests are built once; another table builds again."""
    from d3pm_amd import tables as tbl
    tok = pkg.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=37, schedule="linear", device=DEV)
    Qb_c, Qt_c = Qb.cpu(), Qt.cpu()
    before = tbl.builds
    seen = []
    for _ in range(3):
        seen.append(tbl.protein_tables(Qt_c.to(DEV), Qb_c.to(DEV)))           # fresh device copies, step after step
    assert tbl.builds == before + 1 and seen[0] is seen[1] is seen[2]
    assert tbl.protein_tables(Qt_c, Qb_c, device=DEV) is seen[0]                # the host tensors themselves
    assert tbl.protein_tables(Qt_c.clone(), Qb_c.clone(), device=DEV) is seen[0] and tbl.builds == before + 1
    other = Qb_c.clone()
    other[5] = Qb_c[6]
    assert tbl.protein_tables(Qt_c.to(DEV), other.to(DEV)) is not seen[0] and tbl.builds == before + 2
    # through the loss module, as the notebook calls it
    x0 = torch.randint(0, 22, (4, 50), device=DEV)
    ts = torch.tensor([1, 5, 20, 36], device=DEV)
    pred = torch.randn(4, 50, 22, device=DEV, requires_grad=True)
    lf = pkg.D3PMLVBLoss(tmax=37, tokenizer=tok)
    vals = [float(lf(None, None, pred, x0, None, ts, Qt_c.to(DEV), Qb_c.to(DEV), src=x0).detach()) for _ in range(2)]
    assert vals[0] == vals[1] and tbl.builds == before + 2


def test_generate_replay_golden(pkg):
    g = golden("protein_generate")
    tok = pkg.Tokenizer()
    emb, tvec = T_(g["emb"]), T_(g["tvec"])

    class Toy(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, :]

    expo = T_(g["expo"])
    Qb, Qt = T_(g["Q_bar"]), T_(g["Q_t"])
    sample, strings = pkg.generate_d3pm(Toy(), tok, Qt, Qb, 20, 48, DEV, batch_size=3, x_init=T_(g["x_init"]),
                                        noise_fn=lambda step: expo[step])
    np.testing.assert_array_equal(sample.cpu().numpy(), g["sample"])
    assert strings == [str(s) for s in g["strings"]]
    # tables passed from the host, as the reference's callers do, and internal noise
    s2, st2 = pkg.generate_d3pm(Toy(), tok, Qt.cpu(), Qb.cpu(), 20, 48, DEV, batch_size=5)
    assert s2.shape == (5, 48) and len(st2) == 5 and all(len(s) == 48 for s in st2)
    assert int(s2.min()) >= 0 and int(s2.max()) < 22


def test_reverse_step_against_oracle_full_T(pkg, tables500):
    """One reverse step at several t of the T=500 schedule, BLAT length."""
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    gen = torch.Generator().manual_seed(77)
    Qb_c, Qt_c = Qb.cpu(), Qt.cpu()
    for t in (499, 250, 2, 1):
        x = torch.randint(0, 22, (4, 263), generator=gen)
        logits = torch.randn(4, 263, 24, generator=gen) * 2
        expo = torch.empty(4, 263, 22, dtype=torch.float64).exponential_(1, generator=gen)
        ref = pr.reverse_step(logits, x, t, Qt_c, Qb_c, expo)
        out = reverse_step(tb, logits.to(DEV), x.to(DEV), torch.full((4,), t, device=DEV), expo.to(DEV))
        np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("n_tok", [22, 24, 23])
def test_out_of_range_tokens_and_timesteps_are_clamped(pkg, tables500, n_tok):
    """Tokens outside [0, K-1] and timesteps outside the range an operator accepts index nothing out of bounds: the
    result is exactly that of the values brought to the nearest valid one (n_tok 22 / 24: the token-per-lane
    kernels; 23: the lane-group kernels)."""
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.protein_generation.utils import corrupt_batch
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    gen = torch.Generator().manual_seed(n_tok)
    B, L, K = 7, 150, 22
    x0 = torch.randint(0, K, (B, L), generator=gen)
    xt = torch.randint(0, K, (B, L), generator=gen)
    for a in (x0, xt):
        a.view(-1)[::7] = -2
        a.view(-1)[3::11] = K + 9
    t = torch.tensor([-4, 0, 1, 499, 500, 501, 10 ** 6])
    expo = torch.empty(B, L, K, dtype=torch.float64).exponential_(1, generator=gen).to(DEV)
    pred = torch.randn(B, L, n_tok, generator=gen).to(DEV)
    fix = lambda a: a.clamp(0, K - 1)
    q = torch.rand(B, L, K, dtype=torch.float64, generator=gen).to(DEV)

    def run(x0_, xt_, t_corrupt, t_rev, t_loss):
        x0_, xt_ = x0_.to(DEV), xt_.to(DEV)
        out = [corrupt_batch(tb, x0_, t_corrupt.to(DEV), expo)[0],
               reverse_step(tb, pred, xt_, t_rev.to(DEV), expo),                     # recorded noise
               reverse_step(tb, pred, xt_, t_rev.to(DEV), rng=(11, 0, None))]          # device noise
        p = pred.clone().requires_grad_(True)
        per = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)(None, q, p, x0_, None, t_loss.to(DEV), Qt, Qb,
                                                                         src=xt_)
        per.sum().backward()
        torch.cuda.synchronize()
        return out + [per.detach(), p.grad]
    got = run(x0, xt, t, t, t)
    # t == tmax (500) is a branch of the loss, not an index
    t_loss = torch.where(t == 500, t, t.clamp(1, 499))
    want = run(fix(x0), fix(xt), t.clamp(0, 499), t.clamp(1, 499), t_loss)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"output {i} differs"


def test_non_uniform_tables_fail_loudly(pkg):
    from d3pm_amd.tables import protein_tables
    Q = torch.rand(5, 22, 22, dtype=torch.float64, device=DEV)
    with pytest.raises(NotImplementedError):
        protein_tables(Q, Q)


@pytest.mark.parametrize("n_tok", [24, 22, 23])
def test_reverse_step_device_noise_distribution(pkg, tables500, n_tok):
    """Device-noise mode draws with one uniform per token (inverse CDF): the
    empirical class frequencies must match the oracle's marginal.  Even row
    strides are read as 8-byte pairs, odd ones class by class."""
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    gen = torch.Generator().manual_seed(5)
    for t in (400, 3):
        logits1 = torch.randn(1, n_tok, generator=gen) * 1.5
        # (two identical positions: the oracle's squeeze() needs more than one)
        P = pr.reverse_marginal(logits1.expand(2, n_tok), torch.tensor([7, 7]), Qt.cpu()[t], Qb.cpu()[t - 1])[0].numpy()
        Bn, Ln = 64, 4096
        logits = logits1.expand(Bn, Ln, n_tok).contiguous().to(DEV)
        x = torch.full((Bn, Ln), 7, device=DEV)
        torch.manual_seed(11)
        out = reverse_step(tb, logits, x, torch.full((Bn,), t, device=DEV))
        torch.manual_seed(11)
        again = reverse_step(tb, logits, x, torch.full((Bn,), t, device=DEV))
        assert torch.equal(out, again)
        n = out.numel()
        freq = torch.bincount(out.flatten().cpu(), minlength=22).numpy() / n
        tol = 5 * np.sqrt(P * (1 - P) / n) + 1e-5
        assert (np.abs(freq - P) < tol).all(), (freq, P)
        # the MSA sampler's two knobs: last class down-weighted everywhere, excluded in the first half of the positions
        out = reverse_step(tb, logits, x, torch.full((Bn,), t, device=DEV), last_class_div=1.7, exclude_last_prefix=Ln // 2)
        for part, w_last in ((out[:, :Ln // 2], 0.0), (out[:, Ln // 2:], 1.0 / 1.7)):
            Pm = P.copy()
            Pm[21] *= w_last
            Pm /= Pm.sum()
            n = part.numel()
            freq = torch.bincount(part.flatten().cpu(), minlength=22).numpy() / n
            assert (np.abs(freq - Pm) < 5 * np.sqrt(Pm * (1 - Pm) / n) + 1e-5).all(), (w_last, freq, Pm)


@pytest.mark.parametrize("n_tok,Bn", [(22, 1024), (24, 1024), (22, 1250), (24, 1250)])
def test_reverse_step_device_noise_at_sampler_shape(pkg, tables500, n_tok, Bn):
    """The reverse step at the shapes the sampling drivers run (1024 sequences x 263 positions = 4208 full waves of
    the flat layout, and the benchmark's own sub-batch of 1250 sequences = 5136 waves + a tail of 46 tokens; rows of
    22 or 24 logits): every position has its own logits row and x_t, the sequences are its sample; per-position
    class frequencies against the oracle's marginal."""
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    Ln = 263
    gen = torch.Generator().manual_seed(17 + n_tok)
    for t in (499, 250, 2):
        rows = torch.randn(Ln, n_tok, generator=gen) * 2.0
        xt = torch.randint(0, 22, (Ln,), generator=gen)
        P = pr.reverse_marginal(rows, xt, Qt.cpu()[t], Qb.cpu()[t - 1]).numpy()          # [L, K]
        logits = rows.expand(Bn, Ln, n_tok).contiguous().to(DEV)
        x = xt.expand(Bn, Ln).contiguous().to(DEV)
        tt = torch.full((Bn,), t, device=DEV)
        c = torch.zeros(1, dtype=torch.int64, device=DEV)
        out = reverse_step(tb, logits, x, tt, rng=(3, 0, c))
        assert torch.equal(out, reverse_step(tb, logits, x, tt, rng=(3, 0, c)))           # same stream, same draw
        assert not torch.equal(out, reverse_step(tb, logits, x, tt, rng=(3, 4, c)))
        assert int(out.min()) >= 0 and int(out.max()) < 22
        counts = torch.nn.functional.one_hot(out.cpu(), 22).sum(dim=0).numpy()              # [L, K]
        # exact binomial tails per (position, class) cell -- many cells expect a handful of counts or none, where a
        # normal approximation is useless; 5786 cells x 3 timesteps x 2 row widths at 1e-9 each
        from scipy.stats import binom
        tail = np.minimum(binom.cdf(counts, Bn, P), binom.sf(counts - 1, Bn, P))
        assert tail.min() > 1e-9, (tail.min(), np.unravel_index(tail.argmin(), tail.shape))
        # and pooled over positions, where every class has thousands of counts
        pooled, expect = counts.sum(0) / (Bn * Ln), P.mean(0)
        assert (np.abs(pooled - expect) < 5 * np.sqrt(expect / (Bn * Ln)) + 1e-6).all(), (pooled, expect)
        if t == 499:       # a nearly flat marginal: independent draws never coincide over 263 positions
            assert len({bytes(r) for r in out.cpu().numpy().astype(np.uint8)}) == Bn


@pytest.mark.parametrize("n_tok", [22, 24])
def test_reverse_step_device_noise_mixed_timesteps(pkg, tables500, n_tok):
    """Sequences of 100 positions at three different timesteps: the flat token-per-lane kernel gives a wave 64
    consecutive tokens of the batch, so most waves hold the end of one sequence and the start of the next, each with
    its own timestep; per-timestep class frequencies against the oracle's marginal, and the MSA knobs by position."""
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    gen = torch.Generator().manual_seed(29 + n_tok)
    Bn, Ln = 3 * 700, 100
    ts = (400, 3, 250)
    logits1 = torch.randn(1, n_tok, generator=gen) * 1.5
    logits = logits1.expand(Bn, Ln, n_tok).contiguous().to(DEV)
    x = torch.full((Bn, Ln), 7, device=DEV)
    tt = torch.tensor(ts).repeat(Bn // 3).to(DEV)
    c = torch.zeros(1, dtype=torch.int64, device=DEV)
    out = reverse_step(tb, logits, x, tt, rng=(5, 0, c))
    assert torch.equal(out, reverse_step(tb, logits, x, tt, rng=(5, 0, c)))
    assert int(out.min()) >= 0 and int(out.max()) < 22
    half = reverse_step(tb, logits, x, tt, rng=(6, 0, c), last_class_div=1.7, exclude_last_prefix=Ln // 2)
    for i, t in enumerate(ts):
        P = pr.reverse_marginal(logits1.expand(2, n_tok), torch.tensor([7, 7]), Qt.cpu()[t], Qb.cpu()[t - 1])[0].numpy()
        part = out[i::3]
        n = part.numel()
        freq = torch.bincount(part.flatten().cpu(), minlength=22).numpy() / n
        assert (np.abs(freq - P) < 5 * np.sqrt(P * (1 - P) / n) + 1e-5).all(), (t, freq, P)
        for sl, w_last in ((half[i::3, :Ln // 2], 0.0), (half[i::3, Ln // 2:], 1.0 / 1.7)):
            Pm = P.copy()
            Pm[21] *= w_last
            Pm /= Pm.sum()
            n = sl.numel()
            freq = torch.bincount(sl.flatten().cpu(), minlength=22).numpy() / n
            assert (np.abs(freq - Pm) < 5 * np.sqrt(Pm * (1 - Pm) / n) + 1e-5).all(), (t, w_last, freq, Pm)


@pytest.mark.timeout(600)
def test_graphed_reverse_sampler_full_length(pkg, tables500):
    """The whole T = 500 loop (499 steps) of 1024 sequences of 263 positions as one HIP graph, as the sampling
    benchmark replays it: tokens in range, the same seed and start give the same sequences, replays differ."""
    from d3pm_amd.graphs import GraphedReverseSampler
    from d3pm_amd.tables import protein_tables
    tok, Qb, Qt = tables500
    tb = protein_tables(Qt, Qb)
    Bn, Ln = 1024, 263
    frozen = torch.randn(Bn, Ln, 22, generator=torch.Generator().manual_seed(1)).to(DEV)
    model = lambda x, tt: frozen
    x_init = torch.randint(0, 22, (Bn, Ln), device=DEV)
    a = GraphedReverseSampler(tb, model, Bn, Ln, 500, DEV, seed=21)
    b = GraphedReverseSampler(tb, model, Bn, Ln, 500, DEV, seed=21)
    ya, yb = a.sample(x_init).clone(), b.sample(x_init).clone()
    assert ya.shape == (Bn, Ln) and int(ya.min()) >= 0 and int(ya.max()) < 22
    assert torch.equal(ya, yb)
    ya2 = a.sample(x_init).clone()
    assert not torch.equal(ya, ya2)
    # after 499 steps with frozen logits the tokens follow the last steps' marginal, which favours the classes
    # the logits favour: the most frequent token of a position is among its top logits more often than chance
    assert (ya == frozen.argmax(dim=-1)).double().mean() > 1.5 / 22


def test_generate_msa_replay_golden(pkg):
    from d3pm_amd.protein_generation.generate import generate_msa_d3pm
    g = golden("protein_generate_msa")
    tok = pkg.Tokenizer()
    emb, tvec = T_(g["emb"]), T_(g["tvec"])

    class Toy(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, None, :]

    ea, eq = T_(g["expo_all"]), T_(g["expo_query"])          # [steps, B, D*L, K], [steps, B, L, K-1]
    L = eq.shape[2]
    noise = ea.clone()
    noise[:, :, :L, :21] = eq                                  # query-row rows: the draws of the second multinomial
    Qb, Qt = T_(g["Q_bar"]), T_(g["Q_t"])
    sample, strings = generate_msa_d3pm(Toy(), 2, 3, 16, DEV, timesteps=12, penalty_value=float(g["penalty"]), Q_bar=Qb,
                                        Q=Qt, tokenizer=tok, start_query=False, x_init=T_(g["x_init"]),
                                        noise_fn=lambda step: noise[step])
    np.testing.assert_array_equal(sample.cpu().numpy(), g["sample"])
    assert [s[0] for s in strings] == [str(s) for s in g["strings"]]
    # device noise, pinned query row: the query survives, no gap is drawn into it
    query = torch.randint(0, 21, (16,))
    s2, _ = generate_msa_d3pm(Toy(), 4, 3, 16, DEV, timesteps=12, penalty_value=2.0, Q_bar=Qb, Q=Qt, tokenizer=tok,
                              start_query=True, query_sequence=query)
    assert s2.shape == (4, 3, 16) and torch.equal(s2[:, 0].cpu(), query.expand(4, 16))
    s3, _ = generate_msa_d3pm(Toy(), 64, 3, 16, DEV, timesteps=12, penalty_value=0.0, Q_bar=Qb, Q=Qt, tokenizer=tok,
                              start_query=False)
    assert int((s3[:, 0] == 21).sum()) == 0 and int((s3[:, 1:] == 21).sum()) > 0
    with pytest.raises(ValueError):
        generate_msa_d3pm(Toy(), 1, 3, 16, DEV, timesteps=12, Q_bar=Qb, Q=Qt, tokenizer=tok, start_query=True)


def test_collaters(pkg, tables500):
    from d3pm_amd.protein_generation.dataset import Collater, CollaterMSA
    from d3pm_amd.protein_generation.utils import CollaterV2
    tok, Qb, Qt = tables500
    g = golden("protein_collate_v2")
    seqs = [tok.untokenize(r) for r in g["tokens"]]
    col = CollaterV2(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb)
    for t in (1, 37, 500):
        src, src_oh, ts, tgt, tgt_oh, Q_, Qb_, qx = col(seqs, t, T_(g[f"expo_t{t}"]))
        np.testing.assert_array_equal(src.cpu().numpy(), g[f"src_t{t}"])
        np.testing.assert_array_equal(ts.cpu().numpy(), g[f"timesteps_t{t}"])
        np.testing.assert_allclose(qx.cpu().numpy(), g[f"q_t{t}"], rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(tgt.cpu().numpy(), g["tokens"])
        assert src_oh.dtype == torch.float64 and torch.equal(src_oh.argmax(-1), src) and Q_ is Qt and Qb_ is Qb
    lean = CollaterV2(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb, lean=True, reference_timesteps=False)
    src, ts, tgt = lean(seqs, 500, T_(g["expo_t500"]))
    np.testing.assert_array_equal(src.cpu().numpy(), g["src_t500"])
    assert ts.tolist() == [500] * 4

    # Collater: timesteps from numpy's global RNG in the reference's order, corruption against the oracle
    gen = torch.Generator().manual_seed(5)
    expo = torch.empty(4, len(seqs[0]), 22, dtype=torch.float64).exponential_(1, generator=gen)
    np.random.seed(3)
    want_t = [np.random.randint(1, 500) for _ in seqs]
    np.random.seed(3)
    out = Collater(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb)([(s, f"n{i}") for i, s in enumerate(seqs)], expo)
    src, src_oh, ts, tgt, tgt_oh, _, _, qx, names = out
    assert ts.tolist() == want_t and names == ["n0", "n1", "n2", "n3"]
    Qb_c = Qb.cpu()
    for i in range(4):
        xo, po = pr.corrupt(pr.one_hot(torch.from_numpy(g["tokens"][i])), Qb_c[want_t[i]], expo[i])
        np.testing.assert_array_equal(src[i].cpu().numpy(), xo.numpy())
        np.testing.assert_allclose(qx[i].cpu().numpy(), po.numpy(), rtol=1e-12, atol=1e-15)

    # CollaterMSA: one timestep per MSA; q_x holds every MSA twice, as the reference returns it
    msas = [[seqs[0], seqs[1], seqs[2]], [seqs[3], seqs[0], seqs[1]]]
    expo_m = torch.empty(2, 3, len(seqs[0]), 22, dtype=torch.float64).exponential_(1, generator=gen)
    np.random.seed(4)
    want_tm = [np.random.randint(1, 100) for _ in msas]
    np.random.seed(4)
    Qb100, Qt100 = tok.q_uniform_schedule(num_steps=100, device=DEV)
    src, src_oh, ts, tgt, tgt_oh, _, _, qx = CollaterMSA(tokenizer=tok, num_steps=100, Q=Qt100, Q_bar=Qb100)(msas, expo_m)
    assert src.shape == (2, 3, len(seqs[0])) and qx.shape == (4, 3, len(seqs[0]), 22) and ts.tolist() == want_tm
    assert torch.equal(qx[0], qx[1]) and torch.equal(qx[2], qx[3])
    Qb100_c = Qb100.cpu()
    for i in range(2):
        flat = torch.from_numpy(np.stack([tok.tokenize(s) for s in msas[i]])).flatten()
        xo, po = pr.corrupt(pr.one_hot(flat), Qb100_c[want_tm[i]], expo_m[i].flatten(0, 1))
        np.testing.assert_array_equal(src[i].flatten().cpu().numpy(), xo.numpy())
        np.testing.assert_allclose(qx[2 * i].flatten(0, 1).cpu().numpy(), po.numpy(), rtol=1e-12, atol=1e-15)


def test_collaters_golden(pkg, tables500):
    """Collater and CollaterMSA against the reference's own (dataset.py:196-300, run by tools/make_golden.py with
    its multinomial draws recorded): timesteps in the reference's draw order, tokens bit-exact, probabilities, the
    return tuples' shapes and dtypes, q_x doubled by the MSA collater; then the lean form and the loss on top."""
    from d3pm_amd.protein_generation.dataset import Collater, CollaterMSA
    tok, Qb, Qt = tables500
    g = golden("protein_collaters")
    seqs = [tok.untokenize(r) for r in g["tokens"][:8]]
    np.random.seed(int(g["np_seed"]))
    out = Collater(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb)([(s, f"n{i}") for i, s in enumerate(seqs)],
                                                                 T_(g["expo"]))
    src, src_oh, ts, tgt, tgt_oh, Qr, Qbr, qx, names = out
    assert Qr is Qt and Qbr is Qb and names == [str(n) for n in g["names"]]
    assert src.dtype == torch.int64 and ts.dtype == torch.int64 and src_oh.dtype == torch.float64
    np.testing.assert_array_equal(ts.cpu().numpy(), g["timesteps"])
    np.testing.assert_array_equal(src.cpu().numpy(), g["src"])
    np.testing.assert_array_equal(tgt.cpu().numpy(), g["tokenized"])
    np.testing.assert_allclose(qx.cpu().numpy()[:, :4], g["q_rows"], rtol=1e-12, atol=1e-18)
    assert src_oh.shape == (8, 263, 22) and torch.equal(src_oh.argmax(-1), src) and torch.equal(tgt_oh.argmax(-1), tgt)
    msas = [[tok.untokenize(r[:60]) for r in g["tokens"][4 * m:4 * m + 4]] for m in range(3)]
    np.random.seed(int(g["msa_np_seed"]))
    src, src_oh, ts, tgt, tgt_oh, _, _, qx = CollaterMSA(tokenizer=tok, num_steps=500, Q=Qt, Q_bar=Qb, num_seqs=4)(
        msas, T_(g["msa_expo"]).reshape(3, 4, 60, 22))
    np.testing.assert_array_equal(ts.cpu().numpy(), g["msa_timesteps"])
    np.testing.assert_array_equal(src.cpu().numpy(), g["msa_src"])
    np.testing.assert_array_equal(tgt.cpu().numpy(), g["msa_tokenized"])
    assert tuple(qx.shape) == tuple(g["msa_q_shape"])
    np.testing.assert_allclose(qx.cpu().numpy()[:, :, :3], g["msa_q_rows"], rtol=1e-12, atol=1e-18)


def test_custom_mutant_matrix_golden(pkg, tmp_path):
    from d3pm_amd.protein_generation.utils import custom_mutant_matrix
    g = golden("protein_mutant_scores")
    tok = pkg.Tokenizer()
    focus = str(g["focus_seq"])
    trimmed = [c for c in focus if c == c.upper()]
    L = len(trimmed)
    flat, Em, o = g["expo"], [], 0
    for r in g["expo_rows"].tolist():
        Em.append(torch.from_numpy(flat[o:o + r * L * 22].reshape(r, L, 22)))
        o += r * L * 22
    csv = tmp_path / "mutants.csv"
    csv.write_text(str(g["csv"]))
    emb, tvec = T_(g["emb"]), T_(g["tvec"])

    class Toy(torch.nn.Module):
        def forward(self, x, tt):
            return emb[x] + tvec[tt][:, None, :]

    n_iter, mb, steps = int(g["n_iter"]), int(g["minibatch"]), int(g["steps"])
    nb = (len(g["descriptors"]) + 1 + mb - 1) // mb
    np.random.seed(11)          # the shuffle and the middle timestep come from numpy, as in the reference
    lf = pkg.D3PMLVBLoss(tmax=steps, tokenizer=tok, evaluation=True)
    desc, delta = custom_mutant_matrix(str(csv), Toy(), DEV, focus, str(g["focus_name"]), trimmed, tok, lf, None, 0.0,
                                       steps, T_(g["Q_t"]), T_(g["Q_bar"]), N_pred_iterations=n_iter, minibatch_size=mb,
                                       noise_fn=lambda i, j, h: Em[(i * nb + j // mb) * 3 + h])
    assert desc == [str(d) for d in g["descriptors"]]
    np.testing.assert_allclose(delta, g["delta_elbos"], rtol=1e-5, atol=1e-6)
    # CSV output variant
    np.random.seed(11)
    custom_mutant_matrix(str(csv), Toy(), DEV, focus, str(g["focus_name"]), trimmed, tok, lf, None, 0.0, steps,
                         T_(g["Q_t"]), T_(g["Q_bar"]), N_pred_iterations=1, minibatch_size=mb,
                         filename_prefix=str(tmp_path / "out"))
    lines = (tmp_path / "out_samples-1_elbo_predictions.csv").read_text().splitlines()
    assert [l.split(";")[0] for l in lines] == desc


def test_graphed_reverse_sampler(pkg):
    from d3pm_amd.graphs import GraphedReverseSampler
    from d3pm_amd.protein_generation.generate import reverse_step
    from d3pm_amd.tables import protein_tables
    tok = pkg.Tokenizer()
    T, B, L = 20, 6, 48
    Qb, Qt = tok.q_uniform_schedule(num_steps=T, device=DEV)
    tb = protein_tables(Qt, Qb)
    g = golden("protein_generate")
    emb, tvec = T_(g["emb"]), T_(g["tvec"])
    model = lambda x, tt: emb[x] + tvec[tt][:, None, :]
    smp = GraphedReverseSampler(tb, model, B, L, T, DEV, seed=9)
    x_init = torch.randint(0, 22, (B, L), device=DEV)
    a = smp.sample(x_init).clone()
    counter_a = int(smp.counter.item())
    b = smp.sample(x_init).clone()
    assert a.shape == (B, L) and int(a.min()) >= 0 and int(a.max()) < 22 and not torch.equal(a, b)
    # the replay equals the same loop launched step by step on the same Philox stream
    c = torch.full((1,), counter_a, dtype=torch.int64, device=DEV)
    s = x_init.clone()
    for i, t in enumerate(range(T - 1, 0, -1)):
        tt = torch.full((B,), t, dtype=torch.long, device=DEV)
        s = reverse_step(tb, model(s, tt), s, tt, rng=(9, 4 * i, c))
    assert torch.equal(s, a)


@pytest.mark.parametrize("graph", [True, False])
def test_sample_and_save_sharded_single_rank(pkg, tmp_path, graph):
    """The rank-sharded sampling driver without a process group: 37 sequences in
    sub-batches of 16 (the last one partial), FASTA in the reference's format."""
    from d3pm_amd.protein_generation.generate import sample_and_save_sharded
    tok = pkg.Tokenizer()
    T, L = 12, 21
    Qb, Qt = tok.q_uniform_schedule(num_steps=T, device=DEV)
    emb = torch.randn(22, 24, generator=torch.Generator().manual_seed(3)).to(DEV)

    class Tiny(torch.nn.Module):
        def forward(self, x, t):
            return emb[x] + 0.01 * t[:, None, None].float()

    out = str(tmp_path) + "/"
    strings = sample_and_save_sharded(out, 37, Tiny(), tok, Qt, Qb, T, L, DEV, batch_size=16, seed=5, graph=graph)
    assert len(strings) == 37 and all(len(s) == L and set(s) <= set("ACDEFGHIKLMNPQRSTVWYX-") for s in strings)
    lines = open(out + "generated_samples_string.fasta").read().split("\n")
    assert lines[0] == ">SEQUENCE_0" and lines[1] == strings[0] and lines[72] == ">SEQUENCE_36" and lines[74] == ""
    assert len(set(strings)) > 30          # independent draws
