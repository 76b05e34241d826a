"""GPU parity of the image-style process: HIP kernels (through the facade and
the C ABI) against the golden vectors captured from the reference and against
the CPU oracle on seeded inputs.

Bars: integer outputs (sampled tokens) bit-exact; float32 logits, losses and
gradients within 1e-5 relative (absolute floor stated per check)."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import image_style as im

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-5
# Absolute floor for per-sample losses (bits/dim).  A KL term is a sum of
# differences of log-probabilities of magnitude up to ~14, each carrying float32
# rounding (6e-8 relative), so the reference's own float32 value of a KL that
# is close to zero (late timesteps: 1e-4 and below) is only defined to a few
# 1e-7 absolute.  Losses of ordinary size are held to RTOL.
LOSS_ATOL = 5e-7
# Absolute floor for gradients, as a fraction of the largest gradient element of the tensor.  A gradient element is
# p_j (g_j - sum_i p_i g_i): a difference of terms of the size of the largest element, so float32 rounding leaves
# an absolute error of that scale whatever the element's own size.  The reference's own float32 gradients on the
# golden inputs differ from a float64 evaluation of the same formulas by up to 3.7e-7 of the largest element
# (2e-4 relative on elements above a thousandth of it); test_oracle_golden.py measures that.  Elements are held
# to RTOL above a floor of under three times that figure.
GRAD_FLOOR = 1e-6
# Absolute floor for posterior logits.  An element is log(a + eps) + log(b + eps), two float32 logarithms of
# magnitude up to 13.8 each whose arguments carry one float32 rounding (6e-8 relative, i.e. 6e-8 absolute in the
# logarithm): where the two nearly cancel against nothing -- a logit close to zero -- the value is defined to about
# 2.4e-7 (two argument roundings, two logarithm roundings at 1 ulp of 1e-7 near zero), and the sum's own rounding
# adds at most an ulp of the larger term.  Elements are held to RTOL above 2e-6.
POST_ATOL = 2e-6
# Table entries are probabilities in [0,1] built in float64; entries such as
# diag = 1 - rowsum carry an absolute rounding of a few 1e-16 whatever their size.
TABLE_ATOL = 1e-15


@pytest.fixture(scope="module")
def pkg():
    import d3pm_amd
    return d3pm_amd


def T_(a, dev=DEV):
    return torch.from_numpy(np.array(a)).to(dev)


def make(pkg, g, kind, bits, bands=None, lam=0.001, loss_type="hybrid"):
    return pkg.DiscreteDiffusion(betas=torch.from_numpy(g["betas"]), transition_mat_type=kind, num_bits=bits,
                                 transition_bands=bands, model_prediction="x_start", model_output="logits",
                                 loss_type=loss_type, hybrid_coeff=lam, device=DEV)


def close(a, b, rtol=RTOL, atol=1e-6, what=""):
    a, b = (x.detach().cpu().numpy() if torch.is_tensor(x) else x for x in (a, b))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def check_tokens(got, want, margin, what):
    """Bit-exact, except where the reference's own top-2 margin is below
    float32 resolution of the logits (none in the committed fixtures)."""
    got = got.cpu().numpy()
    bad = got != want
    if bad.any():
        assert (margin[bad] < 1e-5).all(), f"{what}: {bad.sum()} tokens differ with margin >= 1e-5"
    assert bad.sum() == 0, f"{what}: {bad.sum()} near-tie tokens differ (margins {margin[bad]})"


def check_downstream(d, g, own_tables=False):
    """own_tables: the process runs on tables it built itself, which differ from
    the reference's by float64 rounding (1e-16 absolute), so gathered
    probabilities are compared to that accuracy instead of bit for bit."""
    x0, t = T_(g["x0"]), T_(g["t"])
    noise, logits, noise2, w = T_(g["noise"]), T_(g["logits"]), T_(g["noise2"]), T_(g["w"])
    if own_tables:
        np.testing.assert_allclose(d.q_probs(x0, t).cpu().numpy(), g["q_probs"], rtol=1e-6, atol=1e-15)
    else:
        np.testing.assert_array_equal(d.q_probs(x0, t).cpu().numpy(), g["q_probs"])
    x_t = d.q_sample(x0, t, noise)
    assert x_t.dtype == torch.int64 and x_t.shape == x0.shape
    check_tokens(x_t, g["x_t"], g["q_sample_margin"], "q_sample(uniform noise)")
    gum = im.gumbel_from_uniform(torch.from_numpy(g["noise"])).to(DEV)
    check_tokens(d.q_sample(x0, t, gum, noise_kind="gumbel"), g["x_t"], g["q_sample_margin"], "q_sample(gumbel)")
    x_t = T_(g["x_t"])
    close(d.q_posterior_logits(x0, x_t, t, x_start_logits=False), g["post_true"], atol=POST_ATOL, what="post_true")
    close(d.q_posterior_logits(logits, x_t, t, x_start_logits=True), g["post_model"], atol=POST_ATOL, what="post_model")
    ml, px0 = d.p_logits(lambda x, tt: logits, x=x_t, t=t)
    close(ml, g["p_logits"], atol=POST_ATOL, what="p_logits")
    assert px0 is logits
    vb, _ = d.vb_terms_bpd(lambda x, tt: logits, x_start=x0, x_t=x_t, t=t)
    close(vb, g["vb"], atol=LOSS_ATOL, what="vb")
    close(d.cross_entropy_start(x0, logits), g["ce"], atol=1e-7, what="ce")
    for lt in ("hybrid", "kl", "cross_entropy_x_start"):
        d.loss_type = lt
        lg = logits.clone().requires_grad_(True)
        # replay of training_losses with the recorded t and noise
        losses = d.training_losses(lambda x, tt: lg, x_start=x0, t=t, noise=noise)
        close(losses, g["losses_" + lt], atol=LOSS_ATOL, what="losses " + lt)
        (gr,) = torch.autograd.grad((losses * w).sum(), lg)
        ref = g["grad_w_" + lt]
        close(gr, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what="grad " + lt)
    d.loss_type = "hybrid"
    s, probs = d.p_sample(lambda x, tt: logits, x=x_t, t=t, noise=noise2)
    check_tokens(s, g["p_sample"], g["p_sample_margin"], "p_sample")
    close(probs, g["p_sample_probs"], atol=1e-7, what="p_sample probs")
    if g["prior_bpd"].size:
        close(d.prior_bpd(x0), g["prior_bpd"], atol=LOSS_ATOL, what="prior_bpd")


def test_uniform_k32_golden(pkg):
    g = golden("image_uniform_k32")
    d = make(pkg, g, "uniform", 5)
    ts = g["slice_t"]
    np.testing.assert_allclose(d.q_mats.cpu().numpy()[ts], g["q_mats_slices"], rtol=1e-12, atol=TABLE_ATOL)
    np.testing.assert_allclose(d.transpose_q_onestep_mats.cpu().numpy()[ts], g["qT_onestep_slices"], rtol=1e-12, atol=TABLE_ATOL)
    assert d.tables().structured
    check_downstream(d, g)
    # the same through the dense kernels
    d.force_dense = True
    assert not d.tables().structured
    check_downstream(d, g)


def test_uniform_cosine_tables(pkg):
    g = golden("image_uniform_k32_cosine_tables")
    d = make(pkg, g, "uniform", 5)
    np.testing.assert_allclose(d.q_mats.cpu().numpy()[g["slice_t"]], g["q_mats_slices"], rtol=1e-12, atol=TABLE_ATOL)


def test_absorbing_k32_golden(pkg):
    g = golden("image_absorbing_k32")
    d = make(pkg, g, "absorbing", 5)
    ts = g["slice_t"]
    np.testing.assert_allclose(d.q_mats.cpu().numpy()[ts], g["q_mats_slices"], rtol=1e-12, atol=TABLE_ATOL)
    np.testing.assert_allclose(d.transpose_q_onestep_mats.cpu().numpy()[ts], g["qT_onestep_slices"], rtol=1e-12, atol=TABLE_ATOL)
    assert d.tables().structured
    check_downstream(d, g)
    d.force_dense = True
    check_downstream(d, g)


@pytest.mark.parametrize("name,kind,bands", [("gaussian_k16_full", "gaussian", None),
                                             ("gaussian_k16_b5", "gaussian", 5),
                                             ("uniform_k16_b3", "uniform", 3)])
def test_small_dense_tables_golden(pkg, name, kind, bands):
    g = golden("image_" + name)
    d = make(pkg, g, kind, 4, bands)
    np.testing.assert_allclose(d.q_mats.cpu().numpy(), g["q_mats"], rtol=1e-12, atol=TABLE_ATOL)
    np.testing.assert_allclose(d.transpose_q_onestep_mats.cpu().numpy(), g["qT_onestep"], rtol=1e-12, atol=TABLE_ATOL)
    assert not d.tables().structured
    # downstream on the reference's own tables, so that table rounding cannot move a token
    d.q_mats = T_(g["q_mats"])
    d.transpose_q_onestep_mats = T_(g["qT_onestep"])
    check_downstream(d, g)


@pytest.mark.parametrize("dense_matmul", ["f16x2", "f32"])
def test_gaussian_k256_golden(pkg, dense_matmul):
    """The reference's own K = 256 Gaussian run, on both contraction paths: float32 emulated on the float16 matrix
    instructions (the default) and the float32 matrix instructions."""
    g = golden("image_gaussian_k256")
    d = make(pkg, g, "gaussian", 8)
    d.dense_matmul = dense_matmul
    assert d.tables().dense_matmul == dense_matmul
    ts = g["slice_t"]
    np.testing.assert_allclose(d.q_mats.cpu().numpy()[ts], g["q_mats_slices"], rtol=1e-11, atol=TABLE_ATOL)
    np.testing.assert_allclose(d.transpose_q_onestep_mats.cpu().numpy()[ts], g["qT_onestep_slices"], rtol=1e-12,
                               atol=TABLE_ATOL)
    assert not d.tables().structured
    check_downstream(d, g, own_tables=True)


def test_seeded_against_oracle_odd_shapes(pkg):
    """K not a multiple of 4, int32 tokens, 2-D data, B=1 and npos=1; K = 4 is the alphabet no specialised loss kernel
    takes (the lane-group kernel k_hybrid_loss)."""
    for bits, B, shape in ((3, 5, (7,)), (3, 1, (1,)), (6, 3, (5, 2)), (7, 2, (3, 3, 1)), (2, 4, (9,)), (2, 200, (33,))):
        K = 2 ** bits
        betas = pkg.generate_betas(0.02, 0.5, 12, "linear")
        for kind in ("uniform", "absorbing", "gaussian"):
            d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                                      model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                      hybrid_coeff=0.01, device=DEV)
            p = im.build_process(betas, kind, bits, None, hybrid_coeff=0.01)
            gen = torch.Generator().manual_seed(bits * 100 + B)
            # the oracle follows the reference's 4-D data convention
            x0 = torch.randint(0, K, (B,) + shape, generator=gen, dtype=torch.int32)
            x0_4d = x0.reshape(B, -1, 1, 1)
            t = torch.randint(0, 12, (B,), generator=gen, dtype=torch.int32)
            t[0] = 0
            noise = torch.rand(x0_4d.shape + (K,), generator=gen)
            logits = torch.randn(x0_4d.shape + (K,), generator=gen)
            xt_ref = im.forward_sample(p, x0_4d, t, noise)
            lg_ref = logits.clone().requires_grad_(True)
            loss_ref = im.losses_given_xt(p, lg_ref, x0_4d, xt_ref, t)
            loss_ref.sum().backward()
            xt = d.q_sample(x0.to(DEV), t.to(DEV), noise.reshape(x0.shape + (K,)).to(DEV))
            assert torch.equal(xt.cpu().reshape(xt_ref.shape), xt_ref)
            lg = logits.reshape(x0.shape + (K,)).to(DEV).requires_grad_(True)
            loss = d.training_losses(lambda x, tt: lg, x_start=x0.to(DEV), t=t.to(DEV),
                                     noise=noise.reshape(x0.shape + (K,)).to(DEV))
            loss.sum().backward()
            close(loss, loss_ref.detach().numpy(), atol=1e-6, what=f"loss {kind} K={K}")
            ref = lg_ref.grad.numpy().reshape(lg.shape)
            close(lg.grad, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what=f"grad {kind} K={K}")


@pytest.mark.parametrize("kind,bits", [("uniform", 5), ("absorbing", 4), ("gaussian", 4), ("uniform", 3)])
def test_p_logits_gradient_against_oracle(pkg, kind, bits):
    """p_logits / q_posterior_logits(x_start_logits=True) carry autograd as in the reference (:312-381): a loss a
    caller builds on the posterior logits reaches the denoiser output.  Structured tables through the backward
    kernel, dense ones through the library contraction; t = 0 rows pass the gradient through."""
    K, T, B, shape = 2 ** bits, 12, 4, (5, 3)
    betas = pkg.generate_betas(0.02, 0.5, T, "linear")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.01, device=DEV)
    p = im.build_process(betas, kind, bits, None, hybrid_coeff=0.01)
    gen = torch.Generator().manual_seed(bits)
    x_t = torch.randint(0, K, (B,) + shape + (1,), generator=gen)
    t = torch.tensor([0, 1, 5, T - 1])
    z = torch.randn((B,) + shape + (1, K), generator=gen)
    W = torch.randn(z.shape, generator=gen)
    z_ref = z.clone().requires_grad_(True)
    ml_ref = im.model_logits(p, z_ref, x_t, t)[0]
    (ml_ref * W).sum().backward()
    z_dev = z.to(DEV).requires_grad_(True)
    ml, px0 = d.p_logits(lambda x, tt: z_dev * 1.0, x=x_t.to(DEV), t=t.to(DEV))
    close(ml, ml_ref.detach().numpy(), atol=POST_ATOL, what="p_logits")
    (ml * W.to(DEV)).sum().backward()
    ref = z_ref.grad.numpy()
    close(z_dev.grad, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what=f"d p_logits / d logits {kind}")
    post = d.q_posterior_logits(z_dev, x_t.to(DEV), t.to(DEV), x_start_logits=True)
    assert post.requires_grad and torch.equal(post.detach(), ml.detach())
    with torch.no_grad():
        assert not d.q_posterior_logits(z_dev, x_t.to(DEV), t.to(DEV), x_start_logits=True).requires_grad


def test_backward_twice_with_retain_graph(pkg):
    """The loss nodes hand their gradient buffer to the first backward; a second backward through a retained graph
    (the reference supports it) recomputes, with the new upstream gradient."""
    K, T, B, npos = 32, 20, 6, 50
    betas = pkg.generate_betas(0.0, 0.0, T, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type="uniform", num_bits=5, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.01, device=DEV)
    x0 = torch.randint(0, K, (B, npos), device=DEV)
    t = torch.randint(0, T, (B,), device=DEV)
    z = torch.randn(B, npos, K, device=DEV, requires_grad=True)
    noise = torch.rand(B, npos, K, device=DEV)
    losses = d.training_losses(lambda x, tt: z, x_start=x0, t=t, noise=noise)
    w1, w2 = torch.rand(B, device=DEV), torch.rand(B, device=DEV)
    g1, = torch.autograd.grad((losses * w1).sum(), z, retain_graph=True)
    g1 = g1.clone()
    g2, = torch.autograd.grad((losses * w2).sum(), z, retain_graph=True)
    ones, = torch.autograd.grad(losses.sum(), z)
    close(g1, (ones * w1[:, None, None]).cpu().numpy(), rtol=1e-5, atol=1e-9, what="first backward")
    close(g2, (ones * w2[:, None, None]).cpu().numpy(), rtol=1e-5, atol=1e-9, what="second backward")


def _oracle_process(betas, kind, K, lam):
    """`im.build_process` for any K (the reference only builds K = 2 ** num_bits)."""
    one = im._onestep_uniform if kind == "uniform" else im._onestep_absorbing
    steps = [one(betas[t], K, None) if kind == "uniform" else one(betas[t], K) for t in range(betas.shape[0])]
    onestep = torch.stack(steps)
    cur, chain = onestep[0].numpy(), []
    for t in range(betas.shape[0]):
        cur = cur if t == 0 else np.tensordot(cur, onestep[t].numpy(), axes=[[1], [0]])
        chain.append(cur)
    return im.Process(betas=betas, kind=kind, K=K, bands=None, q_mats=torch.tensor(np.stack(chain)),
                      qT_onestep=onestep.permute(0, 2, 1), hybrid_coeff=lam, loss_type="hybrid")


@pytest.mark.parametrize("K", [8, 16, 24, 32])
@pytest.mark.parametrize("B,npos", [(3, 70), (200, 130), (2, 1100), (1, 64), (192, 1100), (300, 70), (600, 300), (450, 40), (257, 513)])
def test_token_per_lane_loss_against_oracle(pkg, K, B, npos):
    """The fused loss kernels that give a lane a whole token (structured tables, K in {8, 16, 24, 32}, rows 16-byte
    aligned) or, for K = 32 with one workgroup per batch element (B >= 192), a pair of lanes: ragged last wave,
    several rounds per workgroup, chunked small batches and one workgroup per batch element, batches beyond one
    element per CU (B = 300, 600: 256-thread workgroups whose waves walk several passes; B = 450: a second round of
    one workgroup per element), t = 0 rows, weighted upstream gradient -- against the CPU oracle; K = 24 through the C ABI on tables passed in (the facade only
    builds K = 2 ** num_bits)."""
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS
    from d3pm_amd.tables import ImageTables
    T, lam = 12, 0.01
    betas = pkg.generate_betas(0.02, 0.5, T, "linear")
    for kind in ("uniform", "absorbing"):
        p = _oracle_process(betas, kind, K, lam)
        tb = ImageTables(p.q_mats.to(DEV), p.qT_onestep.to(DEV), transposed_onestep=True)
        assert tb.structured
        gen = torch.Generator().manual_seed(K * 1000 + B + npos)
        x0 = torch.randint(0, K, (B, npos, 1, 1), generator=gen)
        t = torch.randint(0, T, (B,), generator=gen, dtype=torch.int32)
        t[0] = 0
        noise = torch.rand(B, npos, 1, 1, K, generator=gen)
        logits = (2.0 * torch.randn(B, npos, 1, 1, K, generator=gen)).requires_grad_(True)
        w = torch.rand(B, generator=gen) + 0.5
        xt = im.forward_sample(p, x0, t, noise)
        loss_ref = im.losses_given_xt(p, logits, x0, xt, t)
        (loss_ref * w).sum().backward()
        lg, x0d, xtd, td, wd = (a.detach().to(DEV).contiguous() for a in (logits, x0, xt, t, w))
        vb, ce, loss = (torch.empty(B, device=DEV) for _ in range(3))
        dl = torch.full_like(lg, float("nan"))
        wsb = lib.d3pm_hybrid_loss_workspace(B, npos, K)
        ws = torch.empty(max(wsb // 4, 1), device=DEV)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0d), 8, ptr(xtd), 8, ptr(td), 4, B, npos, LOSS["hybrid"],
                                   lam, 1.0, ptr(wd), ptr(vb), ptr(ce), ptr(loss), ptr(dl), None, None, 0, ptr(ws), wsb,
                                   stream_ptr()))
        torch.cuda.synchronize()
        close(loss, loss_ref.detach().numpy(), atol=1e-6, what=f"loss {kind} K={K}")
        close(vb + lam * ce, loss_ref.detach().numpy(), atol=1e-6, what=f"vb + lam ce {kind} K={K}")
        ref = logits.grad.numpy()
        close(dl, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what=f"grad {kind} K={K}")
        # forward only (no gradient asked for): the kernels' other variant, same losses
        loss_f = torch.empty(B, device=DEV)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0d), 8, ptr(xtd), 8, ptr(td), 4, B, npos, LOSS["hybrid"],
                                   lam, 1.0, None, None, None, ptr(loss_f), None, None, None, 0, ptr(ws), wsb,
                                   stream_ptr()))
        close(loss_f, loss_ref.detach().numpy(), atol=1e-6, what=f"forward-only loss {kind} K={K}")


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,npos", [(3, 70), (200, 130), (2, 1100), (256, 96), (1, 1)])
def test_wide_structured_loss_against_oracle(pkg, K, B, npos):
    """The fused loss on column-structured tables with 64, 128 and 256 classes (k_hybrid_loss_wide: 16 classes a lane,
    K / 16 lanes a token): ragged last passes, chunked batches summed by a second launch, one workgroup per element
    for a caller without a workspace, t = 0 rows, weighted upstream gradient, forward only -- against the CPU
    oracle, uniform and absorbing tables (the absorbing column sits inside one 16-byte piece: the per-class
    logarithms of that piece)."""
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS
    from d3pm_amd.tables import ImageTables
    T, lam = 12, 0.01
    betas = pkg.generate_betas(0.02, 0.5, T, "linear")
    for kind in ("uniform", "absorbing"):
        p = _oracle_process(betas, kind, K, lam)
        tb = ImageTables(p.q_mats.to(DEV), p.qT_onestep.to(DEV), transposed_onestep=True)
        assert tb.structured
        gen = torch.Generator().manual_seed(K * 1000 + B + npos)
        x0 = torch.randint(0, K, (B, npos, 1, 1), generator=gen)
        t = torch.randint(0, T, (B,), generator=gen, dtype=torch.int32)
        t[0] = 0
        noise = torch.rand(B, npos, 1, 1, K, generator=gen)
        logits = (2.0 * torch.randn(B, npos, 1, 1, K, generator=gen)).requires_grad_(True)
        w = torch.rand(B, generator=gen) + 0.5
        xt = im.forward_sample(p, x0, t, noise)
        loss_ref = im.losses_given_xt(p, logits, x0, xt, t)
        (loss_ref * w).sum().backward()
        ref = logits.grad.numpy()
        lg, x0d, xtd, td, wd = (a.detach().to(DEV).contiguous() for a in (logits, x0, xt, t, w))
        wsb = lib.d3pm_hybrid_loss_workspace(B, npos, K)
        ws = torch.empty(max(wsb // 4, 1), device=DEV)
        for workspace in ((ptr(ws), wsb), (None, 0)) if B >= 192 else ((ptr(ws), wsb),):
            vb, ce, loss = (torch.empty(B, device=DEV) for _ in range(3))
            dl = torch.full_like(lg, float("nan"))
            check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0d), 8, ptr(xtd), 8, ptr(td), 4, B, npos,
                                       LOSS["hybrid"], lam, 1.0, ptr(wd), ptr(vb), ptr(ce), ptr(loss), ptr(dl), None, None,
                                       0, workspace[0], workspace[1], stream_ptr()))
            torch.cuda.synchronize()
            close(loss, loss_ref.detach().numpy(), atol=1e-6, what=f"loss {kind} K={K}")
            close(vb + lam * ce, loss_ref.detach().numpy(), atol=1e-6, what=f"vb + lam ce {kind} K={K}")
            close(dl, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what=f"grad {kind} K={K}")
        loss_f = torch.empty(B, device=DEV)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0d), 8, ptr(xtd), 8, ptr(td), 4, B, npos, LOSS["hybrid"],
                                   lam, 1.0, None, None, None, ptr(loss_f), None, None, None, 0, ptr(ws), wsb,
                                   stream_ptr()))
        close(loss_f, loss_ref.detach().numpy(), atol=1e-6, what=f"forward-only loss {kind} K={K}")


@pytest.mark.parametrize("dense_matmul", ["f16x2", "f32"])
@pytest.mark.parametrize("B,L,bits", [(2, 150, 8), (3, 64, 8), (1, 65, 8), (9, 333, 8), (2, 150, 7), (9, 333, 7), (1, 65, 6),
                                      (9, 333, 6)])
def test_k256_matrix_core_tiles_against_oracle(pkg, B, L, bits, dense_matmul):
    """K = 256 runs its contractions on the matrix cores in tiles of 64 tokens:
    several tiles per workgroup, a partial last tile, a single-token tail, a t = 0
    batch element, more batch elements than one round of the XCD mapping.  Both
    contraction paths: 'f16x2' (float32 emulated with two float16 pieces per operand;
    the training loss on the register-tile kernel, everything else on the tile
    kernel) and 'f32' (the float32 matrix instructions).  K = 64 and 128: the register-tile kernel serves the loss
    with and without the gradient ('f16x2'), the vector-ALU tile kernel everything else."""
    K, T = 2 ** bits, 20
    betas = pkg.generate_betas(1e-3, 0.3, T, "linear")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type="gaussian", num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.01, device=DEV)
    d.dense_matmul = dense_matmul
    assert (d.tables().qbar_f16x2 is not None) == (dense_matmul == "f16x2")
    p = im.build_process(betas, "gaussian", bits, None, hybrid_coeff=0.01)
    gen = torch.Generator().manual_seed(1000 * B + L)
    x0 = torch.randint(0, K, (B, L, 1, 1), generator=gen)
    t = torch.randint(1, T, (B,), generator=gen, dtype=torch.int32)
    t[0] = 0
    noise = torch.rand(B, L, 1, 1, K, generator=gen)
    logits = 3 * torch.randn(B, L, 1, 1, K, generator=gen)
    xt_ref = im.forward_sample(p, x0, t, noise)
    for lt in ("hybrid", "kl", "cross_entropy_x_start"):
        p.loss_type = d.loss_type = lt
        lg_ref = logits.clone().requires_grad_(True)
        loss_ref = im.losses_given_xt(p, lg_ref, x0, xt_ref, t)
        w = torch.linspace(0.5, 1.5, B)
        (loss_ref * w).sum().backward()
        lg = logits.to(DEV).requires_grad_(True)
        loss = d.training_losses(lambda x, tt: lg, x_start=x0.to(DEV), t=t.to(DEV), noise=noise.to(DEV))
        (loss * w.to(DEV)).sum().backward()
        close(loss, loss_ref.detach().numpy(), atol=LOSS_ATOL, what=f"loss {lt}")
        ref = lg_ref.grad.numpy()
        close(lg.grad, ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max(), what=f"grad {lt}")
        with torch.no_grad():                                  # forward only: its own kernel instantiation
            loss_f = d.training_losses(lambda x, tt: logits.to(DEV), x_start=x0.to(DEV), t=t.to(DEV), noise=noise.to(DEV))
        close(loss_f, loss_ref.detach().numpy(), atol=LOSS_ATOL, what=f"forward-only loss {lt}")
    d.loss_type = "hybrid"
    post = d.q_posterior_logits(logits.to(DEV), xt_ref.to(DEV), t.to(DEV), x_start_logits=True)
    close(post, im.posterior_logits(p, logits, xt_ref, t, True).numpy(), atol=POST_ATOL, what="posterior logits")
    # device-noise sampling: the same generator state gives the same tokens, t = 0 rows are the argmax of the
    # logits, and the sampled classes follow the posterior (every sampled class has non-negligible probability)
    out = []
    for _ in range(2):
        torch.manual_seed(77)
        out.append(d.p_sample(lambda x, tt: logits.to(DEV), x=xt_ref.to(DEV), t=t.to(DEV))[0])
    assert torch.equal(out[0], out[1])
    assert torch.equal(out[0][0].cpu(), logits[0].argmax(-1))
    lp = torch.log_softmax(post.cpu(), -1).gather(-1, out[0].cpu()[..., None])[..., 0]
    assert float(lp[1:].min()) > -25.0 if B > 1 else True


def _logistic_f64(loc, ls, K, eps=1e-6):
    """The head's formula in float64 (centres as the float32 linspace the reference uses)."""
    loc, ls = loc.double().unsqueeze(-1), ls.double().unsqueeze(-1)
    s = torch.exp(-(ls - 2.0))
    c = torch.linspace(-1.0, 1.0, K).double() - loc
    hw = float(np.float32(0.5 * (2.0 / (K - 1.0))))
    lo = torch.nn.functional.logsigmoid(s * (c - hw))
    hi = torch.nn.functional.logsigmoid(s * (c + hw))
    return hi + torch.log1p(-torch.exp(lo - hi) + float(np.float32(eps)))


def _logistic_condition(loc, ls, K, eps=1e-6):
    """1 + eps - exp(lo - hi), the argument of the final logarithm: one float32
    rounding of exp(lo - hi) (6e-8) moves the logit by 6e-8 / this."""
    loc, ls = loc.double().unsqueeze(-1), ls.double().unsqueeze(-1)
    s = torch.exp(-(ls - 2.0))
    c = torch.linspace(-1.0, 1.0, K).double() - loc
    hw = float(np.float32(0.5 * (2.0 / (K - 1.0))))
    lo = torch.nn.functional.logsigmoid(s * (c - hw))
    hi = torch.nn.functional.logsigmoid(s * (c + hw))
    return (1.0 + eps - torch.exp(lo - hi)).numpy()


def test_logistic_head_golden(pkg):
    """model_output == 'logistic_pars': bin logits and their gradient against the
    reference's recorded float32 values.  Far-tail bins are rounding-dominated in
    the reference itself (log of 1 + eps - exp(lo - hi) with lo - hi -> 0), so the
    bar is: no further from the float64 value of the formula than 3x the
    reference's own float32 result is, plus 1e-6 relative."""
    g = golden("image_misc")
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, 20, "jsd"), transition_mat_type="uniform", num_bits=4,
                              transition_bands=None, model_prediction="x_start", model_output="logistic_pars",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    loc = T_(g["logistic_loc"]).requires_grad_(True)
    ls = T_(g["logistic_log_scale"]).requires_grad_(True)
    out = d._get_logits_from_logistic_pars(loc, ls)
    assert out.shape == loc.shape + (16,) and out.dtype == torch.float32
    (out * T_(g["logistic_w"])).sum().backward()
    l64, s64 = torch.from_numpy(g["logistic_loc"]).double().requires_grad_(True), \
        torch.from_numpy(g["logistic_log_scale"]).double().requires_grad_(True)
    o64 = _logistic_f64(l64, s64, 16)
    (o64 * torch.from_numpy(g["logistic_w"]).double()).sum().backward()
    for mine, ref, exact, what in ((out.detach(), g["logistic_logits"], o64.detach(), "logits"),
                                   (loc.grad, g["logistic_grad_loc"], l64.grad, "d loc"),
                                   (ls.grad, g["logistic_grad_log_scale"], s64.grad, "d log_scale")):
        exact = exact.numpy()
        err_ref = np.abs(ref - exact)
        err = np.abs(mine.cpu().numpy() - exact)
        bound = 3 * err_ref.max() + 1e-6 * np.abs(exact).max()
        assert err.max() <= bound, f"{what}: {err.max():.3e} from float64, reference {err_ref.max():.3e}"
    # and element by element, with the rounding sensitivity of each bin spelled out
    cond = _logistic_condition(torch.from_numpy(g["logistic_loc"]), torch.from_numpy(g["logistic_log_scale"]), 16)
    diff = np.abs(out.detach().cpu().numpy() - g["logistic_logits"])
    assert (diff <= 2e-5 * np.abs(g["logistic_logits"]) + 2e-6 + 4e-7 / cond).all(), diff.max()
    assert (cond > 1e-2).mean() > 0.5        # most bins are held to 4e-5 or better


def test_logistic_head_against_oracle_k256(pkg):
    """K = 256 (four bins per lane), odd position count, wide range of scales."""
    gen = torch.Generator().manual_seed(31)
    loc = torch.rand(3, 7, 5, generator=gen) * 2.4 - 1.2
    ls = torch.randn(3, 7, 5, generator=gen)
    w = torch.randn(3, 7, 5, 256, generator=gen)
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, 20, "jsd"), transition_mat_type="uniform", num_bits=8,
                              transition_bands=None, model_prediction="x_start", model_output="logistic_pars",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    lr, sr = loc.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    ref = im.logits_from_logistic_pars(lr, sr, 256)
    (ref * w).sum().backward()
    l64, s64 = loc.double().requires_grad_(True), ls.double().requires_grad_(True)
    o64 = _logistic_f64(l64, s64, 256)
    (o64 * w.double()).sum().backward()
    lg, sg = loc.to(DEV).requires_grad_(True), ls.to(DEV).requires_grad_(True)
    out = d._get_logits_from_logistic_pars(lg, sg)
    (out * w.to(DEV)).sum().backward()
    for mine, r, exact, what in ((out.detach(), ref.detach(), o64.detach(), "logits"), (lg.grad, lr.grad, l64.grad, "d loc"),
                                 (sg.grad, sr.grad, s64.grad, "d log_scale")):
        exact = exact.numpy()
        err_ref, err = np.abs(r.numpy() - exact), np.abs(mine.cpu().numpy() - exact)
        assert err.max() <= 3 * err_ref.max() + 1e-6 * np.abs(exact).max(), (what, err.max(), err_ref.max())
    # empty input
    e = d._get_logits_from_logistic_pars(torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, device=DEV))
    assert e.shape == (0, 4, 256)


def test_calc_bpd_loop_golden(pkg):
    """calc_bpd_loop with the logistic_pars head, replayed on the reference's
    per-step noise: every timestep's bound term, the prior term and the total."""
    g = golden("image_bpd_loop")
    d = pkg.DiscreteDiffusion(betas=torch.from_numpy(g["betas"]), transition_mat_type="gaussian", num_bits=4,
                              transition_bands=None, model_prediction="x_start", model_output="logistic_pars",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    loc, ls = T_(g["loc"]), T_(g["log_scale"])
    res = d.calc_bpd_loop(lambda x, t: (loc, ls), x_start=T_(g["x0"]), noise=T_(g["noises"]))
    assert res["vbterms"].shape == (3, 40)
    close(res["vbterms"], g["vbterms"], rtol=1e-5, atol=LOSS_ATOL, what="vbterms")
    close(res["prior"], g["prior"], atol=LOSS_ATOL, what="prior")
    close(res["total"], g["total"], rtol=1e-5, atol=40 * LOSS_ATOL, what="total")
    # device noise: a different corruption draw, same expectation within Monte-Carlo spread of 10 positions
    res2 = d.calc_bpd_loop(lambda x, t: (loc, ls), x_start=T_(g["x0"]))
    assert torch.isfinite(res2["total"]).all() and res2["prior"].allclose(res["prior"])


def test_graphed_bpd_sweep(pkg):
    """The whole calc_bpd_loop sweep replayed as one HIP graph equals the eager
    sweep on the same Philox stream; a second replay draws different corruption."""
    from d3pm_amd.graphs import GraphedBpdSweep
    T, K, B = 30, 32, 5
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randint(0, K, (B, 40, 1, 1), generator=gen).to(DEV)
    emb = torch.randn(K, K, generator=gen).to(DEV)
    temb = torch.randn(T, K, generator=gen).to(DEV)
    model = lambda x, t: emb[x] + temb[t.long()][:, None, None, None, :]      # depends on x_t and t
    sweep = GraphedBpdSweep(d, model, x0, seed=11)
    r1 = {k: v.clone() for k, v in sweep.replay().items()}
    counter = torch.full((1,), 4 * T, dtype=torch.int64, device=DEV)           # the counter's value in that replay
    eager = d.calc_bpd_loop(model, x_start=x0, rng=(11, counter))
    for k in ("total", "vbterms", "prior"):
        assert torch.equal(r1[k], eager[k]), k
    r2 = sweep.replay()
    assert not torch.equal(r1["vbterms"], r2["vbterms"]) and torch.equal(r1["prior"], r2["prior"])
    assert r2["vbterms"].shape == (B, T) and torch.isfinite(r2["total"]).all()


@pytest.mark.parametrize("kind,bits", [("gaussian", 8), ("gaussian", 7), ("gaussian", 6), ("uniform", 5), ("absorbing", 5)])
def test_p_sample_device_noise_distribution(pkg, kind, bits):
    """p_sample with noise drawn on the device (one uniform per token, inverse CDF
    over the posterior on the dense path): class frequencies over many identical
    tokens against softmax of the parity-tested posterior logits."""
    K, T, B, L = 2 ** bits, 50, 8, 8192
    betas = pkg.generate_betas(1e-3, 0.2, T, "linear") if kind == "gaussian" else pkg.generate_betas(0, 0, T, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.01, device=DEV)
    gen = torch.Generator().manual_seed(bits)
    row = (2.0 * torch.randn(K, generator=gen)).to(DEV)
    logits = row.expand(B, L, 1, 1, K).contiguous()
    x_t = torch.full((B, L, 1, 1), K // 3, device=DEV)
    if kind == "absorbing":
        x_t[2:] = K // 2          # the mask class, early and late timesteps: the one-uniform draw's extra mass on x_t
    t = torch.tensor([1, 2, 7, 20, 33, 49, 10, 5], dtype=torch.int32, device=DEV)
    torch.manual_seed(123)
    s, _ = d.p_sample(lambda x, tt: logits, x=x_t, t=t)
    post = d.q_posterior_logits(logits[:, :1], x_t[:, :1], t, x_start_logits=True)          # [B,1,1,1,K]
    P = torch.softmax(post.double(), -1).reshape(B, K).cpu().numpy()
    # the sampling loop's call shape (no probabilities requested) takes the hardware exp2 / rcp softmax
    s_loop, none = d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False)
    assert none is None
    for draws in (s, s_loop):
        for b in range(B):
            freq = torch.bincount(draws[b].flatten().cpu(), minlength=K).numpy() / L
            tol = 5 * np.sqrt(P[b] * (1 - P[b]) / L) + 2e-4
            assert (np.abs(freq - P[b]) < tol).all(), (b, int(t[b]), np.abs(freq - P[b]).max())


@pytest.mark.parametrize("kind,bits", [("uniform", 8), ("absorbing", 8), ("uniform", 7), ("absorbing", 6)])
def test_p_sample_device_noise_structured_wide_alphabets(pkg, kind, bits):
    """The sampling loop's call (device noise, no probabilities out) on structured tables with 64 / 128 / 256
    classes, where a token sits on 8 or 16 lanes: class frequencies over many identical tokens against the softmax
    of the parity-tested posterior logits (a different timestep per batch element), the same draws under the same
    generator state, and the noise-free argmax of the logits at t = 0 on random rows."""
    K, T, B, L = 2 ** bits, 50, 8, 16384
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, T, "jsd"), transition_mat_type=kind, num_bits=bits,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    assert d.tables().structured
    gen = torch.Generator().manual_seed(40 + bits)
    row = (2.0 * torch.randn(K, generator=gen)).to(DEV)
    logits = row.expand(B, L, 1, 1, K).contiguous()
    x_t = torch.full((B, L, 1, 1), K // 3, device=DEV)
    if kind == "absorbing":
        x_t[4:] = K // 2                                           # the mask class
    t = torch.tensor([1, 2, 7, 20, 33, 49, 10, 5], dtype=torch.int32, device=DEV)
    post = d.q_posterior_logits(logits[:, :1], x_t[:, :1], t, x_start_logits=True)
    P = torch.softmax(post.double(), -1).reshape(B, K).cpu().numpy()
    out = []
    for _ in range(2):
        torch.manual_seed(9)
        out.append(d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False)[0])
    assert torch.equal(out[0], out[1])
    assert int(out[0].min()) >= 0 and int(out[0].max()) < K
    for b in range(B):
        freq = torch.bincount(out[0][b].flatten().cpu(), minlength=K).numpy() / L
        tol = 5 * np.sqrt(P[b] * (1 - P[b]) / L) + 2e-4
        assert (np.abs(freq - P[b]) < tol).all(), (b, int(t[b]), np.abs(freq - P[b]).max())
    # t = 0: argmax of the logits, first maximum, ragged position count
    lg = torch.randn(3, 77, 1, 1, K, generator=gen).to(DEV)
    lg[1, 5, 0, 0, 7] = lg[1, 5, 0, 0, K - 2] = 50.0               # a tie: the first index wins
    xz = torch.randint(0, K, (3, 77, 1, 1), generator=gen).to(DEV)
    tz = torch.tensor([0, 0, 3], dtype=torch.int32, device=DEV)
    sz = d._p_sample_from_logits(lg, xz, tz, None, "uniform", want_probs=False)[0]
    assert torch.equal(sz[:2].cpu(), lg[:2].cpu().argmax(-1))
    assert int(sz[1, 5, 0, 0]) == 7


@pytest.mark.parametrize("bits", [6, 7, 8])
def test_p_sample_device_noise_dense_register_tile(pkg, bits):
    """The sampling loop's call on dense Gaussian tables with 64 / 128 / 256 classes (register-tile kernel on the float16
    split tables): a different row, x_t and timestep per token, ragged position count.  Per class, the count of draws
    against the sum of the parity-tested posterior probabilities over the tokens (5 sigma of the Poisson-binomial);
    t = 0 elements are the argmax of the logits (first index on ties); the same generator state repeats the draws."""
    K, T, B, L = 2 ** bits, 50, 12, 1000 + 37
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(1e-3, 0.2, T, "linear"), transition_mat_type="gaussian",
                              num_bits=bits, transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    assert d.tables().dense_matmul == "f16x2" and not d.tables().structured
    gen = torch.Generator().manual_seed(70 + bits)
    logits = (2.0 * torch.randn(B, L, 1, 1, K, generator=gen)).to(DEV)
    logits[1, 5, 0, 0, 7] = logits[1, 5, 0, 0, K - 2] = 50.0
    logits[0, L - 1, 0, 0, 17] = logits[0, L - 1, 0, 0, 6] = 60.0       # a tie whose first index sits on the later lane
    x_t = torch.randint(0, K, (B, L, 1, 1), generator=gen).to(DEV)
    t = torch.tensor([0, 0, 1, 2, 3, 7, 12, 20, 33, 41, 49, 5], dtype=torch.int32, device=DEV)
    out = []
    for _ in range(2):
        torch.manual_seed(11)
        out.append(d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False)[0])
    s = out[0]
    assert torch.equal(out[0], out[1])
    assert int(s.min()) >= 0 and int(s.max()) < K
    assert torch.equal(s[:2].cpu(), logits[:2].cpu().argmax(-1))
    assert int(s[1, 5, 0, 0]) == 7 and int(s[0, L - 1, 0, 0]) == 6
    post = d.q_posterior_logits(logits[2:], x_t[2:], t[2:], x_start_logits=True)
    P = torch.softmax(post.double(), -1).reshape(-1, K)
    cnt = torch.bincount(s[2:].flatten(), minlength=K).double()
    z = (cnt - P.sum(0)) / (P * (1 - P)).sum(0).clamp_min(1e-12).sqrt()
    assert float(z.abs().max()) < 5, z
    # and the draws follow each token's own distribution: mean log-probability of the drawn classes against its
    # expectation (minus the entropy), 5 sigma
    lp = torch.log(P.clamp_min(1e-300))
    got = lp.gather(1, s[2:].reshape(-1, 1)).sum()
    mean = (P * lp).sum()
    var = ((P * lp * lp).sum(1) - (P * lp).sum(1) ** 2).sum()
    assert abs(float(got - mean)) < 5 * float(var.sqrt()), (float(got), float(mean), float(var.sqrt()))


@pytest.mark.parametrize("bits", [8, 7, 6])
def test_split_contraction_against_float32_path_full_size(pkg, bits):
    """Dense Gaussian tables at the bench's size (64 x 1024 tokens, two rounds of workgroups on every CU): the
    training loss and its gradient on the float16 split tables (register-tile kernel) against the same call on the
    float32 tables (dense_matmul='f32': float32 matrix instructions at 256 classes, vector ALUs below), both
    parity-tested against the oracle at small sizes.  Every element, 2e-5 relative with a floor of 1e-6 of the
    largest gradient; three draws of the inputs.  Repeated launches (loss, gradient, sampled tokens) are bit-identical:
    a schedule of the contraction whose LDS reads overtook matrix instructions once passed every small case and
    changed 1 - 3 % of the gradient elements from launch to launch at this size."""
    from d3pm_amd.tables import ImageTables
    K, T, B, L = 2 ** bits, 40, 64, 1024
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(1e-3, 0.2, T, "linear"), transition_mat_type="gaussian",
                              num_bits=bits, transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    tb16 = d.tables()
    assert tb16.dense_matmul == "f16x2"
    tb32 = ImageTables(d.q_mats, d.transpose_q_onestep_mats, eps=d.eps, transposed_onestep=True, dense_matmul="f32")
    lib, ptr, check, sp = pkg._lib.lib, pkg._lib.ptr, pkg._lib.check, pkg._lib.stream_ptr()
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(max(wsb // 4, 1), device=DEV)
    for seed in range(3):
        gen = torch.Generator().manual_seed(100 * bits + seed)
        x0 = torch.randint(0, K, (B, L, 1, 1), generator=gen).to(DEV)
        t = torch.randint(0, T, (B,), generator=gen).to(torch.int32).to(DEV)
        xt = d.q_sample(x0, t)
        logits = (2.0 * torch.randn(B, L, 1, 1, K, generator=gen)).to(DEV)
        res = []
        for tb in (tb16, tb32, tb16, tb16):
            loss = torch.empty(B, device=DEV)
            dl = torch.full_like(logits, float("nan"))
            check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                       pkg._lib.LOSS["hybrid"], 0.01, 1.0 / B, None, None, None, ptr(loss), ptr(dl),
                                       None, None, 0, ptr(ws), wsb, sp))
            res.append((loss, dl))
        torch.cuda.synchronize()
        (l16, g16), (l32, g32) = res[:2]
        assert torch.isfinite(g16).all()
        torch.testing.assert_close(l16, l32, rtol=2e-5, atol=0)
        torch.testing.assert_close(g16, g32, rtol=2e-5, atol=1e-6 * float(g32.abs().max()))
        # a launch's result does not depend on timing: repeated launches are bit-identical
        for l, g in res[2:]:
            assert torch.equal(l, l16) and torch.equal(g, g16)
        # forward only (another kernel at 256 classes), and the reverse step with probabilities out
        fw = []
        for tb in (tb16, tb32, tb16):
            loss = torch.empty(B, device=DEV)
            check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                       pkg._lib.LOSS["hybrid"], 0.01, 1.0 / B, None, None, None, ptr(loss), None,
                                       None, None, 0, ptr(ws), wsb, sp))
            fw.append(loss)
        torch.cuda.synchronize()
        torch.testing.assert_close(fw[0], fw[1], rtol=2e-5, atol=0)
        torch.testing.assert_close(fw[0], l16, rtol=2e-6, atol=0)
        assert torch.equal(fw[0], fw[2])
        if seed == 0:
            pr = []
            for tb in (tb16, tb32, tb16):
                smp = torch.empty(B, L, 1, 1, dtype=torch.int64, device=DEV)
                probs = torch.full_like(logits, float("nan"))
                check(lib.d3pm_p_sample(tb.desc(), ptr(logits), K, ptr(xt), 8, ptr(t), 4, B, L, None,
                                        pkg._lib.NOISE_PHILOX, 5, 0, None, ptr(smp), ptr(probs), sp))
                pr.append((smp, probs))
            torch.cuda.synchronize()
            torch.testing.assert_close(pr[0][1], pr[1][1], rtol=2e-5, atol=1e-9)
            assert torch.equal(pr[0][1], pr[2][1]) and torch.equal(pr[0][0], pr[2][0])
            del pr, probs
        # the sampling loop's reverse step on the same tables: the same stream draws the same tokens, launch after launch
        draws = []
        for _ in range(3):
            smp = torch.empty(B, L, 1, 1, dtype=torch.int64, device=DEV)
            check(lib.d3pm_p_sample(tb16.desc(), ptr(logits), K, ptr(xt), 8, ptr(t), 4, B, L, None,
                                    pkg._lib.NOISE_PHILOX, 5, 0, None, ptr(smp), None, sp))
            draws.append(smp)
        torch.cuda.synchronize()
        assert torch.equal(draws[0], draws[1]) and torch.equal(draws[0], draws[2])
        assert int(draws[0].min()) >= 0 and int(draws[0].max()) < K


@pytest.mark.parametrize("kind,bits", [("uniform", 5), ("absorbing", 5), ("uniform", 8), ("absorbing", 7)])
def test_structured_loss_repeatable_at_bench_size(pkg, kind, bits):
    """The fused loss on structured tables at the bench's size (256 x 512 tokens): three launches on the same inputs
    give bit-identical per-element losses and gradients (nothing in a launch depends on timing), and so does the
    launch that also draws the next batch's timesteps and corruption from a fixed stream."""
    K, T, B, L = 2 ** bits, 500, 256, 512
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type=kind, num_bits=bits,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    tb = d.tables()
    assert tb.structured
    lib, ptr, check, sp = pkg._lib.lib, pkg._lib.ptr, pkg._lib.check, pkg._lib.stream_ptr()
    gen = torch.Generator().manual_seed(bits)
    x0 = torch.randint(0, K, (B, L, 1, 1), generator=gen).to(DEV)
    t = torch.randint(0, T, (B,), generator=gen).to(torch.int32).to(DEV)
    xt = d.q_sample(x0, t)
    logits = torch.randn(B, L, 1, 1, K, generator=gen).to(DEV)
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(max(wsb // 4, 1), device=DEV)
    res = []
    for _ in range(3):
        loss = torch.empty(B, device=DEV)
        dl = torch.full_like(logits, float("nan"))
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                   pkg._lib.LOSS["hybrid"], 0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl), None,
                                   None, 0, ptr(ws), wsb, sp))
        res.append((loss, dl))
    torch.cuda.synchronize()
    assert torch.isfinite(res[0][1]).all()
    for l, g in res[1:]:
        assert torch.equal(l, res[0][0]) and torch.equal(g, res[0][1])
    if K == 32:
        nxt = []
        for _ in range(3):
            loss = torch.empty(B, device=DEV)
            dl = torch.full_like(logits, float("nan"))
            t_nx = torch.empty(B, dtype=torch.int32, device=DEV)
            xt_nx = torch.empty(B, L, 1, 1, dtype=torch.int64, device=DEV)
            check(lib.d3pm_hybrid_loss_then_q_sample(
                tb.desc(), ptr(logits), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L, pkg._lib.LOSS["hybrid"], 0.001,
                1.0 / B, None, None, None, ptr(loss), ptr(dl), ptr(ws), wsb, ptr(x0), 8, 3, 0, None, ptr(t_nx),
                ptr(xt_nx), sp))
            nxt.append((loss, dl, t_nx, xt_nx))
        torch.cuda.synchronize()
        assert torch.equal(nxt[0][1], res[0][1]) and torch.equal(nxt[0][0], res[0][0])
        for r in nxt[1:]:
            assert all(torch.equal(a, b) for a, b in zip(r, nxt[0]))


def test_raw_ctypes_binding_as_documented(pkg):
    """The binding a maintainer of the reference would write (plain ctypes, no
    helper of this package): digests from the float64 tables, then the fused loss
    and its gradient; equal to the package's own result."""
    import ctypes as C
    lib = C.CDLL(pkg._lib.LIB_PATH)

    class TablesF32(C.Structure):
        _fields_ = [("K", C.c_int), ("T", C.c_int), ("structured", C.c_int), ("eps", C.c_float),
                    ("qbar_digest", C.c_void_p), ("qone_digest", C.c_void_p),
                    ("qbar", C.c_void_p), ("qoneT", C.c_void_p), ("qbarT", C.c_void_p),
                    ("qbar_f16x2", C.c_void_p), ("qbarT_f16x2", C.c_void_p)]

    K, T, B, npos = 32, 500, 256, 40
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    q_mats = d.q_mats.to(DEV)
    qone = d.transpose_q_onestep_mats.to(DEV).permute(0, 2, 1)
    dg_bar, dg_one = torch.empty(T, 8, K, device=DEV), torch.empty(T, 8, K, device=DEV)
    viol = torch.zeros(2, dtype=torch.int32, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    for tab, dg, off in ((q_mats, dg_bar, 0), (qone, dg_one, 4)):
        rc = lib.d3pm_digest_f32(C.c_void_p(tab.data_ptr()), *map(C.c_long, tab.stride()), T, K, C.c_float(1e-6),
                                 C.c_void_p(dg.data_ptr()), C.c_void_p(viol.data_ptr() + off), C.c_void_p(s))
        assert rc == 0
    assert viol.tolist() == [0, 0]
    tb = TablesF32(K, T, 1, 1e-6, dg_bar.data_ptr(), dg_one.data_ptr(), None, None, None, None, None)
    gen = torch.Generator().manual_seed(4)
    x0 = torch.randint(0, K, (B, npos, 1, 1), generator=gen).to(DEV)
    t = torch.randint(0, T, (B,), generator=gen, dtype=torch.int32).to(DEV)
    x_t = d.q_sample(x0, t)
    logits = torch.randn(B, npos, 1, 1, K, generator=gen).to(DEV)
    loss, dlogits = torch.empty(B, device=DEV), torch.empty_like(logits)
    lib.d3pm_last_error.restype = C.c_char_p
    rc = lib.d3pm_hybrid_loss(C.byref(tb), C.c_void_p(logits.data_ptr()), C.c_long(K),
                              C.c_void_p(x0.data_ptr()), 8, C.c_void_p(x_t.data_ptr()), 8, C.c_void_p(t.data_ptr()), 4,
                              B, C.c_long(npos), 2, C.c_float(0.001), C.c_float(1.0 / B), None,
                              None, None, C.c_void_p(loss.data_ptr()), C.c_void_p(dlogits.data_ptr()),
                              None, None, C.c_uint64(0), None, C.c_size_t(0), C.c_void_p(s))
    assert rc == 0, lib.d3pm_last_error()
    lg = logits.clone().requires_grad_(True)
    ref = d._fused(lg, x0, x_t, t, "hybrid")
    ref.mean().backward()
    assert torch.equal(loss, ref) and torch.equal(dlogits, lg.grad)


@pytest.mark.parametrize("kind,bits", [("uniform", 5), ("gaussian", 8), ("absorbing", 4)])
def test_graphed_p_sample_loop(pkg, kind, bits):
    """The whole p_sample_loop replayed as one HIP graph equals the eager loop on
    the same Philox stream and start tokens; the next replay draws different noise."""
    from d3pm_amd.graphs import GraphedPSampleLoop
    K, T, shape = 2 ** bits, 12, (3, 6, 5, 1)
    betas = pkg.generate_betas(1e-3, 0.3, T, "linear") if kind == "gaussian" else pkg.generate_betas(0, 0, T, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.001, device=DEV)
    gen = torch.Generator().manual_seed(bits)
    emb = torch.randn(K, K, generator=gen).to(DEV)
    temb = torch.randn(T, K, generator=gen).to(DEV)
    model = lambda x, t: emb[x.long()] + temb[t.long()][:, None, None, None, :]
    loop = GraphedPSampleLoop(d, model, shape, seed=21)
    x_init = torch.randint(0, K, shape, generator=gen).to(DEV)
    r1 = loop.sample(x_init).clone()
    counter = torch.full((1,), 4 * T, dtype=torch.int64, device=DEV)      # the counter's value in that replay
    eager = d.p_sample_loop(model, shape=shape, x_init=x_init, rng=(21, counter))
    assert torch.equal(r1, eager) and r1.dtype == torch.int64 and r1.shape == shape
    assert int(r1.min()) >= 0 and int(r1.max()) < K
    r2 = loop.sample(x_init).clone()
    assert not torch.equal(r1, r2)
    r3 = loop.sample()                                  # x_T drawn by the sampler
    assert r3.shape == shape


@pytest.mark.parametrize("kind,bits", [("uniform", 5), ("uniform", 3), ("gaussian", 4), ("gaussian", 8)])
def test_out_of_range_tokens_and_timesteps_are_clamped(pkg, kind, bits):
    """Tokens outside [0, K-1] and timesteps outside [0, T-1] index nothing out of bounds: every operator returns
    exactly what it returns for the values brought to the nearest valid one (uniform 5 bits: the token-per-lane loss
    kernel; 3 bits: the lane-group kernels; gaussian: dense tables, 8 bits on the matrix cores)."""
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS
    K, T = 2 ** bits, 20
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.02, 0.3, T, "linear"), transition_mat_type=kind,
                              num_bits=bits, transition_bands=None, model_prediction="x_start",
                              model_output="logits", loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    gen = torch.Generator().manual_seed(5 + bits)
    B, npos = 6, 96
    x0 = torch.randint(0, K, (B, npos, 1, 1), generator=gen)
    xt = torch.randint(0, K, (B, npos, 1, 1), generator=gen)
    for a in (x0, xt):
        a.view(-1)[::7] = -3
        a.view(-1)[3::11] = K + 5
    t = torch.tensor([-1, 0, T - 1, T, T + 100, 3], dtype=torch.int64)
    noise = torch.rand(B, npos, 1, 1, K, generator=gen)
    logits = torch.randn(B, npos, 1, 1, K, generator=gen)
    fix = lambda a: a.clamp(0, K - 1)

    def run(x0_, xt_, t_):
        x0_, xt_, t_, nz, lg = (a.to(DEV) for a in (x0_, xt_, t_, noise, logits))
        res = [d.q_probs(x0_, t_), d.q_sample(x0_, t_, nz), d.q_posterior_logits(x0_, xt_, t_, x_start_logits=False),
               d.q_posterior_logits(lg, xt_, t_, x_start_logits=True), d.prior_bpd(x0_),
               d.vb_terms_bpd(lambda x, tt: lg, x_start=x0_, x_t=xt_, t=t_)[0],
               d.p_sample(lambda x, tt: lg, x=xt_, t=t_, noise=nz)[0]]
        # the fused loss and its gradient, through the C ABI
        tb = d.tables()
        loss, dl = torch.empty(B, device=DEV), torch.empty_like(lg)
        wsb = lib.d3pm_hybrid_loss_workspace(B, npos, K)
        ws = torch.empty(max(wsb // 4, 1), device=DEV)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0_), 8, ptr(xt_), 8, ptr(t_), 8, B, npos, LOSS["hybrid"],
                                   0.01, 1.0, None, None, None, ptr(loss), ptr(dl), None, None, 0, ptr(ws), wsb,
                                   stream_ptr()))
        torch.cuda.synchronize()
        return res + [loss, dl]
    got = run(x0, xt, t)
    want = run(fix(x0), fix(xt), t.clamp(0, T - 1))
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"output {i} differs"


def test_empty_batch_and_errors(pkg):
    betas = pkg.generate_betas(0, 0, 10, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type="uniform", num_bits=3, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid",
                              hybrid_coeff=0.001, device=DEV)
    x0 = torch.zeros((0, 4, 1, 1), dtype=torch.int64, device=DEV)
    t = torch.zeros((0,), dtype=torch.int32, device=DEV)
    assert d.q_sample(x0, t).shape == (0, 4, 1, 1)
    assert d.q_probs(x0, t).shape == (0, 4, 1, 1, 8)
    with pytest.raises(AssertionError):
        d.q_sample(torch.zeros((2, 4), dtype=torch.int64, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV),
                   torch.rand(2, 4, 7, device=DEV))
    d.loss_type = "nope"
    with pytest.raises(NotImplementedError):
        d.training_losses(lambda x, tt: None, x_start=torch.zeros((2, 4), dtype=torch.int64, device=DEV))
    d.model_output = "xyz"
    d.loss_type = "hybrid"
    with pytest.raises(NotImplementedError):
        d.p_logits(lambda x, tt: None, x=torch.zeros((2, 4), dtype=torch.int64, device=DEV),
                   t=torch.zeros(2, dtype=torch.int32, device=DEV))


def test_full_size_properties(pkg):
    """The benchmark configuration (B=256, L=512, K=32, T=500): properties that
    hold at any size, plus a slice checked against the oracle."""
    B, L, K, T = 256, 512, 32, 500
    betas = pkg.generate_betas(0, 0, T, "jsd")
    torch.manual_seed(3)
    for kind in ("uniform", "absorbing"):
        d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=5, transition_bands=None,
                                  model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                  hybrid_coeff=0.001, device=DEV)
        x0 = torch.randint(0, K, (B, L, 1, 1), device=DEV)
        t = torch.randint(0, T, (B,), dtype=torch.int32, device=DEV)
        t[:3] = torch.tensor([0, 1, T - 1], dtype=torch.int32)
        noise = torch.rand(B, L, 1, 1, K, device=DEV)
        x_t = d.q_sample(x0, t, noise)
        assert int(x_t.min()) >= 0 and int(x_t.max()) < K
        # same noise -> same tokens; Philox noise is repeatable under a seed
        assert torch.equal(x_t, d.q_sample(x0, t, noise))
        torch.manual_seed(9); a = d.q_sample(x0, t)
        torch.manual_seed(9); b = d.q_sample(x0, t)
        assert torch.equal(a, b)
        # t = 0 for uniform jsd has beta_0 = 1/T: most tokens survive
        assert (x_t[0] == x0[0]).float().mean() > 0.9
        lg = torch.randn(B, L, 1, 1, K, device=DEV, requires_grad=True)
        vb, _ = d.vb_terms_bpd(lambda x, tt: lg, x_start=x0, x_t=x_t, t=t)
        assert bool((vb >= -1e-6).all()), "KL terms must be non-negative"
        loss = d.training_losses(lambda x, tt: lg, x_start=x0, t=t, noise=noise)
        loss.mean().backward()
        g = lg.grad
        # gradients of softmax-parameterised losses sum to zero over classes
        assert float(g.sum(-1).abs().max()) < 1e-6 * float(g.abs().max()) * K + 1e-12
        # a prediction that is one-hot on x0 drives the KL to (almost) zero
        sharp = torch.nn.functional.one_hot(x0, K).float() * 60.0
        vb0, _ = d.vb_terms_bpd(lambda x, tt: sharp, x_start=x0, x_t=x_t, t=t)
        assert float(vb0.max()) < 1e-3
        # oracle on a slice
        sl = slice(0, 6)
        p = im.build_process(betas, kind, 5, None, hybrid_coeff=0.001)
        lg_ref = lg.detach()[sl].cpu().clone().requires_grad_(True)
        xt_ref = im.forward_sample(p, x0[sl].cpu(), t[sl].cpu(), noise[sl].cpu())
        assert torch.equal(xt_ref, x_t[sl].cpu())
        loss_ref = im.losses_given_xt(p, lg_ref, x0[sl].cpu(), xt_ref, t[sl].cpu())
        close(loss[sl], loss_ref.detach().numpy(), atol=LOSS_ATOL)
        (loss_ref.sum() / B).backward()
        ref = lg_ref.grad.numpy()
        close(g[sl], ref, rtol=RTOL, atol=GRAD_FLOOR * np.abs(ref).max())


def test_q_sample_device_noise_distribution(pkg):
    """With noise drawn on the device q_sample uses one uniform per token
    (inverse CDF); its empirical distribution must be softmax(log(q + eps))."""
    K, T = 32, 500
    betas = pkg.generate_betas(0, 0, T, "jsd")
    for kind in ("uniform", "absorbing"):
        d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=5, transition_bands=None,
                                  model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                  hybrid_coeff=0.001, device=DEV)
        for tval, x0val in ((3, 5), (250, 16), (499, 31), (40, 0)):
            B, L = 64, 8192
            x0 = torch.full((B, L), x0val, device=DEV)
            t = torch.full((B,), tval, dtype=torch.int32, device=DEV)
            torch.manual_seed(21)
            xt = d.q_sample(x0, t)
            n = xt.numel()
            freq = torch.bincount(xt.flatten().cpu(), minlength=K).numpy() / n
            q = d.q_probs(x0[:1, :1], t[:1]).cpu().numpy().reshape(K).astype(np.float64) + 1e-6
            q /= q.sum()
            tol = 5 * np.sqrt(q * (1 - q) / n) + 2e-6
            assert (np.abs(freq - q) < tol).all(), (kind, tval, freq, q)


@pytest.mark.parametrize("kind,bits,B,L", [("uniform", 5, 256, 96), ("uniform", 5, 6, 700), ("absorbing", 4, 200, 33),
                                            ("gaussian", 4, 5, 40)])
def test_training_step_matches_separate_calls(pkg, kind, bits, B, L):
    """The two-launch step (timestep draw + corruption; loss + batch mean + gradient) against the
    reference-shaped calls on the timesteps and noisy tokens it produced."""
    K, T = 2 ** bits, 50
    betas = pkg.generate_betas(1e-4, 0.02, T, "linear") if kind == "gaussian" else pkg.generate_betas(0.0, 0.0, T, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid", hybrid_coeff=0.01,
                              device=DEV)
    gen = torch.Generator().manual_seed(B + L)
    x0 = torch.randint(0, K, (B, L, 1, 1), generator=gen).to(DEV)
    logits = (torch.randn(B, L, 1, 1, K, generator=gen) * 2).to(DEV).requires_grad_(True)
    counter = torch.full((1,), 40, dtype=torch.int64, device=DEV)
    seen = {}

    def model_fn(x, t):
        seen["x"], seen["t"] = x, t
        return logits * 1.0                       # a non-leaf, so the gradient flows through autograd

    out = d.training_step(model_fn, x_start=x0, rng=(123, counter))
    assert int(counter.item()) == 44              # advanced by the loss kernel
    t, x_t = out["t"], out["x_t"]
    assert t.dtype == torch.int32 and t.shape == (B,) and int(t.min()) >= 0 and int(t.max()) < T
    assert seen["x"] is x_t and seen["t"] is t
    # the same Philox stream through q_sample gives the same corruption
    c2 = torch.full((1,), 40, dtype=torch.int64, device=DEV)
    assert torch.equal(d.q_sample(x0, t, rng=(123, c2)), x_t)
    ref_logits = logits.detach().clone().requires_grad_(True)
    ref = d._fused(ref_logits, x0, x_t, t, "hybrid")
    ref.mean().backward()
    assert torch.equal(out["losses"], ref)
    close(out["loss"], ref.mean(), atol=LOSS_ATOL, what="batch mean")
    close(logits.grad, ref_logits.grad, atol=GRAD_FLOOR * float(ref_logits.grad.abs().max()), what="gradient")
    # second step: fresh draw, the arrival counter was left clean
    logits.grad = None
    out2 = d.training_step(model_fn, x_start=x0, rng=(123, counter))
    assert int(counter.item()) == 48 and not torch.equal(out2["x_t"], x_t)
    close(out2["loss"], out2["losses"].mean(), atol=LOSS_ATOL, what="batch mean, second step")
    # torch's generator as the source, forward only
    torch.manual_seed(7)
    a = d.training_step(lambda x, tt: logits.detach(), x_start=x0, backward=False)
    torch.manual_seed(7)
    b = d.training_step(lambda x, tt: logits.detach(), x_start=x0, backward=False)
    assert torch.equal(a["x_t"], b["x_t"]) and torch.equal(a["t"], b["t"]) and torch.equal(a["losses"], b["losses"])


def test_training_step_timestep_distribution_and_graph(pkg):
    K, T, B, L = 32, 500, 4096, 8
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    x0 = torch.randint(0, K, (B, L, 1, 1), device=DEV)
    logits = torch.randn(B, L, 1, 1, K, device=DEV, requires_grad=True)
    g = pkg.graphs.GraphedTrainingStep(d, lambda x, tt: logits, x0, seed=5, params=[logits])
    ts = []
    for _ in range(8):
        out = g.replay()
        ts.append(out["t"].clone())
        assert logits.grad is not None and torch.isfinite(logits.grad).all()
    ts = torch.cat(ts).cpu().numpy()
    assert len(set(map(bytes, np.split(ts, 8)))) == 8                  # every replay draws anew
    counts = np.bincount(ts, minlength=T)
    exp = len(ts) / T
    chi2 = float(((counts - exp) ** 2 / exp).sum())
    assert counts.min() > 0 and chi2 < T + 6 * np.sqrt(2 * T), chi2    # uniform on [0, T): chi-square within 6 sigma
    ref = d._fused(logits.detach(), x0, out["x_t"], out["t"], "hybrid")
    assert torch.equal(out["losses"], ref)
    np.testing.assert_allclose(float(out["loss"]), float(ref.double().mean()), rtol=1e-5)     # reduced when read
    # on the device: accumulated with one atomic per batch element, equal to the mean up to summation order
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    dm = d.training_step(lambda x, tt: logits, x_start=x0, rng=(9, counter), device_mean=True)
    assert "loss" in dm and dm["loss"].shape == ()
    np.testing.assert_allclose(float(dm["loss"]), float(dm["losses"].double().mean()), rtol=1e-5)


def test_training_step_graph_dense_k256(pkg):
    """The captured training step on dense tables with 256 classes (the register-tile loss kernel with its LDS-DMA
    table stream, the dense forward corruption): every replay draws anew, and its losses and gradient equal the eager
    fused loss on the replay's own x_t and t."""
    K, T, B, L = 256, 30, 10, 200
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(1e-3, 0.2, T, "linear"), transition_mat_type="gaussian", num_bits=8,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    assert d.tables().dense_matmul == "f16x2"
    x0 = torch.randint(0, K, (B, L, 1, 1), device=DEV)
    logits = torch.randn(B, L, 1, 1, K, device=DEV, requires_grad=True)
    g = pkg.graphs.GraphedTrainingStep(d, lambda x, tt: logits, x0, seed=5, params=[logits])
    seen = []
    for _ in range(3):
        out = g.replay()
        seen.append(out["x_t"].clone())
        lg = logits.detach().clone().requires_grad_(True)
        ref = d._fused(lg, x0, out["x_t"], out["t"], "hybrid")
        ref.mean().backward()
        assert torch.equal(out["losses"], ref.detach())
        assert torch.equal(logits.grad, lg.grad)
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


def test_training_step_several_steps_per_graph(pkg):
    """Three consecutive steps captured in one graph over a ring of three batches
    equal three eager steps on the same Philox counter values."""
    K, T, B, L = 32, 500, 64, 48
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    ring = [torch.randint(0, K, (B, L, 1, 1), device=DEV) for _ in range(3)]
    logits = torch.randn(B, L, 1, 1, K, device=DEV, requires_grad=True)
    fn = lambda x, tt: logits
    g = pkg.graphs.GraphedTrainingStep(d, fn, ring, seed=5, params=[logits], steps=3, warmup=1)
    start = int(g.counter.item())                   # the warm-up step advanced it
    g.replay()
    assert int(g.counter.item()) == start + 12      # three steps of four Philox offsets each
    grad_graph = logits.grad.clone()
    counter = torch.full((1,), start, dtype=torch.int64, device=DEV)
    for i in range(3):
        logits.grad = None
        e = d.training_step(fn, x_start=ring[i], rng=(5, counter))
        for k in ("t", "x_t", "losses"):
            assert torch.equal(g.outs[i][k], e[k]), (i, k)
        np.testing.assert_allclose(float(g.outs[i]["loss"]), float(e["loss"]), rtol=1e-5)
    assert torch.equal(grad_graph, logits.grad)      # the last step's gradient
    assert not torch.equal(g.outs[0]["t"], g.outs[1]["t"])


@pytest.mark.parametrize("kind,bits,B,L", [("uniform", 5, 256, 512), ("absorbing", 5, 200, 96), ("uniform", 5, 64, 48),
                                            ("uniform", 4, 256, 128), ("gaussian", 4, 5, 40), ("uniform", 5, 192, 700)])
def test_pipelined_steps_equal_unpipelined(pkg, kind, bits, B, L):
    """A graph whose loss launch of step i also draws the timesteps and noisy tokens of step i + 1
    (d3pm_hybrid_loss_then_q_sample: one launch at K = 32 with a workgroup per batch element, two launches for the
    other shapes) against the eager two-launch steps on the same Philox counter values: timesteps, tokens, losses and
    the last gradient bit for bit, over two replays."""
    K, T, n = 2 ** bits, 500, 4
    betas = pkg.generate_betas(1e-4, 0.02, T, "linear") if kind == "gaussian" else pkg.generate_betas(0.0, 0.0, T, "jsd")
    d = pkg.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                              model_prediction="x_start", model_output="logits", loss_type="hybrid", hybrid_coeff=0.001,
                              device=DEV)
    gen = torch.Generator().manual_seed(B + L)
    ring = [torch.randint(0, K, (B, L, 1, 1), generator=gen).to(DEV) for _ in range(n)]
    logits = (torch.randn(B, L, 1, 1, K, generator=gen) * 2).to(DEV).requires_grad_(True)
    fn = lambda x, tt: logits
    g = pkg.graphs.GraphedTrainingStep(d, fn, ring, seed=5, params=[logits], steps=n, warmup=1, pipelined=True)
    assert g.pipelined
    start = int(g.counter.item())
    assert start == 4 * n                            # the warm-up sequence advanced it once, by 4 n
    counter = torch.full((1,), start, dtype=torch.int64, device=DEV)
    static_grad = None
    for rep in range(2):
        g.replay()
        assert int(g.counter.item()) == start + 4 * n * (rep + 1)
        got = [{k: o[k].clone() for k in ("t", "x_t", "losses")} for o in g.outs]
        if static_grad is None:
            static_grad = logits.grad               # the buffer the captured last step writes on every replay
        grad_graph = static_grad.clone()
        for i in range(n):
            logits.grad = None
            e = d.training_step(fn, x_start=ring[i], rng=(5, counter))
            for k in ("t", "x_t", "losses"):
                assert torch.equal(got[i][k], e[k]), (rep, i, k)
        assert torch.equal(grad_graph, logits.grad)
        assert not torch.equal(got[0]["t"], got[1]["t"])
    # eager use with torch's generator as the noise source: the next batch's corruption comes back under 'next'
    torch.manual_seed(3)
    c0 = d.corrupt_step(x_start=ring[0])
    out = d.loss_step(fn, x_start=ring[0], corrupted=c0, prefetch={"x_start": ring[1]})
    nxt = out["next"]
    assert nxt["x_t"].shape == ring[1].shape and nxt["t"].shape == (B,)
    assert int(nxt["x_t"].min()) >= 0 and int(nxt["x_t"].max()) < K and int(nxt["t"].max()) < T
    ref = d._fused(logits.detach(), ring[0], c0["x_t"], c0["t"], "hybrid")
    assert torch.equal(out["losses"], ref)
    out2 = d.loss_step(fn, x_start=ring[1], corrupted=nxt)
    assert torch.equal(out2["losses"], d._fused(logits.detach(), ring[1], nxt["x_t"], nxt["t"], "hybrid"))


@pytest.mark.parametrize("kind,bits", [("uniform", 5), ("absorbing", 4), ("uniform", 8)])
def test_digest_sampling_rows(pkg, kind, bits):
    """Rows 4-6 of the float32 digest, which the device-noise q_sample kernels copy: w = off + eps, its prefix sums
    in class order (float32, sequential), and P(x_t == x0 = j) = (diag_j + eps) / (sum w - w_j + diag_j + eps) as a
    32-bit integer threshold formed in float64."""
    K, T = 2 ** bits, 60
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type=kind, num_bits=bits,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    tb = d.tables()
    assert tb.structured and tb.qbar_digest.shape == (T, 8, K)
    dg = tb.qbar_digest.cpu().numpy()
    eps = np.float32(1e-6)
    qm = d.q_mats.cpu().numpy()
    for t in (0, 1, T // 2, T - 1):
        diag = np.diagonal(qm[t]).astype(np.float32)
        off = np.array([qm[t][1 if j == 0 else 0, j] for j in range(K)], dtype=np.float32)
        np.testing.assert_array_equal(dg[t, 0], diag)
        np.testing.assert_array_equal(dg[t, 1], off)
        w = off + eps
        np.testing.assert_array_equal(dg[t, 4], w)
        run, cdf = np.float32(0), []
        for j in range(K):
            run = np.float32(run + w[j])
            cdf.append(run)
        np.testing.assert_array_equal(dg[t, 5], np.array(cdf, dtype=np.float32))
        dd = (diag + eps).astype(np.float64)
        stay = dd / (np.float64(cdf[-1]) - w.astype(np.float64) + dd)
        want = np.where(stay >= 1.0, 0xFFFFFFFF, np.minimum(stay * 4294967296.0, 4294967295.0)).astype(np.uint64)
        got = dg[t, 6].view(np.uint32).astype(np.uint64)
        assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1, (t, got[:4], want[:4])


def test_then_q_sample_refuses_aliased_outputs(pkg):
    """d3pm_hybrid_loss_then_q_sample reads this step's x_t / t while it writes the next step's: the same buffers for
    both are refused before anything is launched."""
    from d3pm_amd._lib import lib, ptr, stream_ptr, LOSS
    K, T, B, L = 32, 50, 4, 40
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    tb = d.tables()
    x0 = torch.randint(0, K, (B, L), device=DEV)
    t = torch.randint(0, T, (B,), dtype=torch.int32, device=DEV)
    xt = d.q_sample(x0.view(B, L, 1, 1), t).view(B, L)
    lg = torch.randn(B, L, K, device=DEV)
    dl, loss = torch.empty_like(lg), torch.empty(B, device=DEV)
    wsb = lib.d3pm_hybrid_loss_workspace(B, L, K)
    ws = torch.empty(max(wsb // 4, 1), device=DEV)
    t2, xt2 = torch.empty_like(t), torch.empty_like(xt)

    def call(next_t, next_xt):
        return lib.d3pm_hybrid_loss_then_q_sample(tb.desc(), ptr(lg), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, L,
                                                  LOSS["hybrid"], 0.001, 1.0 / B, None, None, None, ptr(loss), ptr(dl),
                                                  ptr(ws), wsb, ptr(x0), 8, 5, 0, None, ptr(next_t), ptr(next_xt),
                                                  stream_ptr())
    assert call(t2, xt) != 0 and b"must not be" in lib.d3pm_last_error()
    assert call(t, xt2) != 0
    assert call(t2, xt2) == 0
    torch.cuda.synchronize()
    assert int(xt2.min()) >= 0 and int(xt2.max()) < K and int(t2.min()) >= 0 and int(t2.max()) < T


def test_p_sample_flat_mixed_timesteps(pkg):
    """The sampling loop's p_sample at K = 32 (device noise, no probabilities out) takes the flat token-per-lane
    kernel: 64 consecutive tokens of the batch per wave, so with 100 positions per batch element most waves hold
    tokens of two batch elements with different timesteps (one of them t = 0: argmax of the logits).  Class
    frequencies per timestep against softmax of the parity-tested posterior logits; same stream, same draw."""
    K, T, L = 32, 50, 100
    ts = [0, 1, 7, 33, 49]
    B = len(ts) * 400
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0, 0, T, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    gen = torch.Generator().manual_seed(8)
    row = (2.0 * torch.randn(K, generator=gen)).to(DEV)
    logits = row.expand(B, L, 1, 1, K).contiguous()
    x_t = torch.full((B, L, 1, 1), 11, device=DEV)
    t = torch.tensor(ts, dtype=torch.int32).repeat(B // len(ts)).to(DEV)
    c = torch.zeros(1, dtype=torch.int64, device=DEV)
    s, none = d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False, rng=(4, c, 0))
    again, _ = d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False, rng=(4, c, 0))
    assert none is None and torch.equal(s, again)
    other, _ = d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False, rng=(4, c, 4))
    assert not torch.equal(s, other)
    post = d.q_posterior_logits(logits[:len(ts), :1], x_t[:len(ts), :1], t[:len(ts)], x_start_logits=True)
    P = torch.softmax(post.double(), -1).reshape(len(ts), K).cpu().numpy()
    for i, tv in enumerate(ts):
        part = s[i::len(ts)]
        n = part.numel()
        freq = torch.bincount(part.flatten().cpu(), minlength=K).numpy() / n
        if tv == 0:
            assert freq[int(row.argmax())] == 1.0
        else:
            assert (np.abs(freq - P[i]) < 5 * np.sqrt(P[i] * (1 - P[i]) / n) + 2e-5).all(), (tv, np.abs(freq - P[i]).max())


@pytest.mark.parametrize("bits", [8, 4])
def test_dense_q_sample_device_noise_distribution(pkg, bits):
    """q_sample on dense (Gaussian) tables with noise drawn on the device: one uniform per token, inverse CDF over
    q + eps; class frequencies against softmax(log(q + eps)).  K = 256 takes the 16-lane layout."""
    K, T = 2 ** bits, 100
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(1e-3, 0.2, T, "linear"), transition_mat_type="gaussian",
                              num_bits=bits, transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    assert not d.tables().structured
    for tval, x0val in ((2, K // 3), (40, 0), (99, K - 1)):
        B, L = 32, 8192
        x0 = torch.full((B, L), x0val, device=DEV)
        t = torch.full((B,), tval, dtype=torch.int32, device=DEV)
        torch.manual_seed(21)
        xt = d.q_sample(x0, t)
        torch.manual_seed(21)
        assert torch.equal(xt, d.q_sample(x0, t))
        n = xt.numel()
        freq = torch.bincount(xt.flatten().cpu(), minlength=K).numpy() / n
        q = d.q_probs(x0[:1, :1], t[:1]).cpu().numpy().reshape(K).astype(np.float64) + 1e-6
        q /= q.sum()
        assert (np.abs(freq - q) < 5 * np.sqrt(q * (1 - q) / n) + 2e-6).all(), (bits, tval, np.abs(freq - q).max())



def test_operators_reject_mis_shaped_calls(pkg):
    """torch.ops.d3pm_mi355x is a public boundary of its own: a call whose tensors do not describe one batch must
    raise before a pointer reaches the library (one case per operator family), and a well-formed call works without
    the caller making the device current."""
    from d3pm_amd._ops import ops
    from d3pm_amd.tables import protein_tables
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, 20, "jsd"), transition_mat_type="uniform", num_bits=3,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    tb = d.tables()
    B, L, K = 4, 6, 8
    x0 = torch.randint(0, K, (B, L), device=DEV)
    t = torch.randint(0, 20, (B,), device=DEV)
    lg = torch.randn(B, L, K, device=DEV)
    ok = ops.hybrid_loss(*tb.op_args(), lg, x0, x0, t, 2, 0.001, 0.25, None, True)
    assert ok[0].shape == (B,) and ok[3].shape == lg.shape
    bad_t, bad_x, bad_lg = t[:3].contiguous(), x0[:, :5].contiguous(), lg[:, :5].contiguous()
    cases = [
        lambda: ops.q_probs(*tb.op_args(), x0, bad_t),
        lambda: ops.q_sample(*tb.op_args(), x0, t, torch.rand(B, L, K - 1, device=DEV), 0, 0, 0, None),
        lambda: ops.q_sample_draw_t(*tb.op_args(), x0, 0, 0, torch.zeros(2, dtype=torch.int64, device=DEV), None),
        lambda: ops.q_posterior_logits(*tb.op_args(), bad_x, False, x0, t),
        lambda: ops.q_posterior_logits(*tb.op_args(), bad_lg, True, x0, t),
        lambda: ops.q_posterior_logits_backward(*tb.op_args(), lg, t, bad_lg),
        lambda: ops.hybrid_loss(*tb.op_args(), bad_lg, x0, x0, t, 2, 0.001, 0.25, None, True),
        lambda: ops.hybrid_loss(*tb.op_args(), lg, x0, bad_x, t, 2, 0.001, 0.25, None, True),
        lambda: ops.hybrid_loss(*tb.op_args(), lg, x0, x0, t, 2, 0.001, 0.25, torch.ones(B + 1, device=DEV), True),
        lambda: ops.hybrid_loss_step(*tb.op_args(), lg, x0, x0, bad_t, 2, 0.001, 0.25, True, None, None, 0),
        lambda: ops.hybrid_loss_then_q_sample(*tb.op_args(), lg, x0, x0, t, 2, 0.001, 0.25, True, bad_x, 0, 0, None),
        lambda: ops.hybrid_loss_static_(*tb.op_args(), lg, x0, x0, t, 2, 0.001, 0.25, torch.empty(B - 1, device=DEV), None),
        lambda: ops.scale_rows_(torch.empty(7, device=DEV), torch.ones(2, device=DEV), 1.0),
        lambda: ops.p_sample(*tb.op_args(), lg, x0, bad_t, None, 2, 0, 0, None, False),
        lambda: ops.logistic_logits(torch.zeros(5, device=DEV), torch.zeros(4, device=DEV), torch.linspace(-1, 1, 8).to(DEV), 1e-6),
        lambda: ops.logistic_logits_backward(torch.zeros(5, device=DEV), torch.zeros(5, device=DEV),
                                             torch.linspace(-1, 1, 8).to(DEV), torch.zeros(5, 7, device=DEV), 1e-6),
        lambda: ops.digest_f32(torch.zeros(3, 4, 5, dtype=torch.float64, device=DEV), 1e-6),
        lambda: ops.cumprod(torch.zeros(3, 4, dtype=torch.float64, device=DEV)),
        # a float32 digest with the four rows of an older layout: read past its end if it were accepted
        lambda: ops.q_probs([tb.qbar_digest[:, :4].contiguous(), tb.qone_digest] + list(tb.op_args()[0][2:]), K, 20, True,
                            1e-6, x0, t),
    ]
    tok = pkg.Tokenizer()
    Qb, Qt = tok.q_uniform_schedule(num_steps=30, device=DEV)
    pt = protein_tables(Qt, Qb)
    xp = torch.randint(0, 22, (B, L), device=DEV)
    tp = torch.randint(1, 30, (B,), device=DEV)
    pp = torch.randn(B, L, 24, device=DEV)
    assert ops.lvb_loss(*pt.op_args(), pp, xp, xp, tp, None, 30, 0, 0.25, True)[0].shape == (B,)
    cases += [
        lambda: ops.lvb_loss(*pt.op_args(), pp[:, :5].contiguous(), xp, xp, tp, None, 30, 0, 0.25, True),
        lambda: ops.lvb_loss(*pt.op_args(), pp, xp, xp, tp, torch.zeros(B, L, 21, dtype=torch.float64, device=DEV), 30, 0, 0.25, True),
        lambda: ops.lvb_hybrid_loss(*pt.op_args(), pp, xp, xp[:, :5].contiguous(), tp, None, 30, 0, 0.25, 0.1, True),
        lambda: ops.lvb_reverse_step(*pt.op_args(), pp, xp, tp[:2].contiguous(), None, 0, 0, None, 1.0, 0),
        lambda: ops.corrupt(*pt.op_args(), xp, tp[:2].contiguous(), None, 0, 0, False),
        lambda: ops.corrupt(pt.qbar_digest[:, :2].contiguous(), pt.qone_digest, pt.lvb_digest, 22, 30, xp, tp, None, 0, 0, False),
        lambda: ops.lvb_loss(pt.qbar_digest, pt.qone_digest, pt.lvb_digest[:5].contiguous(), 22, 30, pp, xp, xp, tp, None, 30, 0, 0.25, True),
    ]
    for i, call in enumerate(cases):
        with pytest.raises((ValueError, RuntimeError)):
            call()
            pytest.fail(f"mis-shaped call {i} was accepted")



def test_tables_with_a_diagonal_below_the_off_diagonal_are_not_structured(pkg):
    """The one-uniform samplers read a column's diagonal surplus over its off-diagonal value as a non-negative extra
    mass, so a column-structured table whose diagonal is SMALLER than the off-diagonal value must not take the O(K)
    path: the digest counts it as a break and the dense kernels serve it (p_sample frequencies against the
    posterior)."""
    from d3pm_amd.tables import ImageTables
    K, T = 8, 6
    off = torch.full((T, 1, K), 0.13, dtype=torch.float64)
    Q = off.expand(T, K, K).clone()
    idx = torch.arange(K)
    Q[:, idx, idx] = 1.0 - 0.13 * (K - 1)                   # 0.09 < 0.13: rows still sum to one
    bars = [Q[0]]
    for i in range(1, T):
        bars.append(bars[-1] @ Q[i])
    Qbar = torch.stack(bars)
    tb = ImageTables(Qbar.to(DEV), Q.to(DEV))
    assert tb.violations[1] > 0 and not tb.structured
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, T, "jsd"), transition_mat_type="uniform", num_bits=3,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    d.q_mats, d.transpose_q_onestep_mats = Qbar.to(DEV), Q.to(DEV).permute(0, 2, 1)
    assert not d.tables().structured
    B, L = 4, 20000
    row = torch.tensor([0.3, -1.0, 2.0, 0.0, 1.0, -0.5, 0.7, 0.2], device=DEV)
    logits = row.expand(B, L, 1, 1, K).contiguous()
    x_t = torch.full((B, L, 1, 1), 2, device=DEV)
    t = torch.tensor([1, 2, 4, 5], dtype=torch.int32, device=DEV)
    torch.manual_seed(3)
    s, _ = d._p_sample_from_logits(logits, x_t, t, None, "uniform", want_probs=False)
    post = d.q_posterior_logits(logits[:, :1], x_t[:, :1], t, x_start_logits=True)
    P = torch.softmax(post.double(), -1).reshape(B, K).cpu().numpy()
    for b in range(B):
        freq = torch.bincount(s[b].flatten().cpu(), minlength=K).numpy() / L
        assert (np.abs(freq - P[b]) < 5 * np.sqrt(P[b] * (1 - P[b]) / L) + 2e-4).all(), (b, freq, P[b])


def test_training_step_halves_refuse_two_streams(pkg):
    """corrupt_step and loss_step share a Philox counter that the loss launch advances: issued on two streams the
    read and the write are unordered, so the second half refuses a stream other than the first half's."""
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(0.0, 0.0, 20, "jsd"), transition_mat_type="uniform", num_bits=5,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.001, device=DEV)
    x0 = torch.randint(0, 32, (4, 70, 1, 1), device=DEV)
    logits = torch.randn(4, 70, 1, 1, 32, device=DEV, requires_grad=True)
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    c = d.corrupt_step(x_start=x0, rng=(7, counter))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with pytest.raises(RuntimeError, match="one stream"):
            d.loss_step(lambda x, tt: logits, x_start=x0, corrupted=c)
    torch.cuda.current_stream().wait_stream(side)
    out = d.loss_step(lambda x, tt: logits, x_start=x0, corrupted=c)            # the stream of the first half: fine
    assert torch.isfinite(out["losses"]).all() and int(counter.item()) == 4


def test_table_split_f16x2_image(pkg):
    """d3pm_table_split_f16x2: the two float16 planes are hi = float16(2^14 x) and lo = float16(2^14 x - hi), round to
    nearest; together they hold an entry to 2^-22 of its size or 2^-39, whichever is larger; and the element order is
    the documented operand image [T, K/32, K/16, 2, 64, 8] (lane kg << 4 | mm of tile mt, slot s = reduction index
    32 u + 16 (s >> 2) + 4 kg + (s & 3), output class 16 mt + mm)."""
    from d3pm_amd._ops import ops
    T, K = 3, 64
    gen = torch.Generator().manual_seed(5)
    tab = (torch.rand(T, K, K, generator=gen) * torch.logspace(-30, 0, K)[None, None, :]).float()
    tab[0, 0, 0] = 0.0
    tab[1, 3, 5] = 1.0
    img = ops.table_split_f16x2(tab.to(DEV)).cpu()
    assert img.shape == (T, K // 32, K // 16, 2, 64, 8) and img.dtype == torch.float16
    u, mt, lane, sl = torch.meshgrid(torch.arange(K // 32), torch.arange(K // 16), torch.arange(64), torch.arange(8),
                                     indexing="ij")
    k = 32 * u + 16 * (sl >> 2) + 4 * (lane >> 4) + (sl & 3)
    m = 16 * mt + (lane & 15)
    want = tab[:, k, m] * 16384.0                              # [T, K/32, K/16, 64, 8]; exact: a power of two
    hi = want.half()
    lo = (want - hi.float()).half()
    assert torch.equal(img[:, :, :, 0], hi) and torch.equal(img[:, :, :, 1], lo)
    total = (img[:, :, :, 0].double() + img[:, :, :, 1].double()) / 16384.0
    err = (total - tab[:, k, m].double()).abs()
    assert bool((err <= torch.maximum(tab[:, k, m].double() * 2.0 ** -22, torch.tensor(2.0 ** -39, dtype=torch.float64))).all())
    with pytest.raises(ValueError):
        ops.table_split_f16x2(torch.zeros(2, 24, 24, device=DEV))


def test_split_tables_only_for_probabilities(pkg):
    """Tables with entries above 1 (not transition probabilities) stay on the float32 matrix instructions: the float16
    pieces carry 2^14 x the entry."""
    from d3pm_amd.tables import ImageTables
    T, K = 2, 256
    gen = torch.Generator().manual_seed(3)
    q = torch.rand(T, K, K, generator=gen, dtype=torch.float64)
    q = q / q.sum(-1, keepdim=True)
    assert ImageTables(q.to(DEV), q.to(DEV)).dense_matmul == "f16x2"
    big = q.clone()
    big[1, 7, 9] = 3.0
    assert ImageTables(big.to(DEV), q.to(DEV)).dense_matmul == "f32"


def test_split_tables_shape_contract(pkg):
    """The operators refuse float16 split tables of the wrong size, a lone one, or ones attached to another alphabet."""
    from d3pm_amd._ops import ops
    d = pkg.DiscreteDiffusion(betas=pkg.generate_betas(1e-3, 0.3, 6, "linear"), transition_mat_type="gaussian", num_bits=8,
                              transition_bands=None, model_prediction="x_start", model_output="logits",
                              loss_type="hybrid", hybrid_coeff=0.01, device=DEV)
    tabs, K, T, structured, eps = d.tables().op_args()
    assert len(tabs) == 7
    x0 = torch.zeros(2, 5, dtype=torch.int64, device=DEV)
    t = torch.ones(2, dtype=torch.int32, device=DEV)
    ops.q_probs(tabs, K, T, structured, eps, x0, t)
    empty = torch.empty(0, dtype=torch.float16, device=DEV)
    for bad in (tabs[:5] + [tabs[5][:-1], tabs[6]], tabs[:5] + [tabs[5], empty], tabs[:6]):
        with pytest.raises((ValueError, RuntimeError)):
            ops.q_probs(bad, K, T, structured, eps, x0, t)
