"""`d3pm_hybrid_loss_static`, the entry point that allocates nothing (partial sums in a static device array of the
library), against `d3pm_hybrid_loss` with a caller workspace: same kernels, so every output must be bit-identical.
Two calls in a row per configuration, the second on a smaller problem (the static array keeps the first call's
contents).  Runs last: it must pass alone in a fresh process and after every other test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(d, B, npos, seed):
    import d3pm_amd  # noqa: F401
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS
    dev = d.device
    K, T = d.num_pixel_vals, d.num_timesteps
    g = torch.Generator(device="cpu").manual_seed(seed)
    x0 = torch.randint(0, K, (B, npos), generator=g).to(dev)
    t = torch.randint(0, T, (B,), generator=g, dtype=torch.int32).to(dev)
    t[0] = 0
    xt = d.q_sample(x0, t, torch.rand(B, npos, K, generator=g).to(dev))
    lg = torch.randn(B, npos, K, generator=g).to(dev)
    go = torch.rand(B, generator=g).to(dev)
    tb = d.tables()
    outs = []
    for static in (False, True):
        vb, ce, loss = (torch.full((B,), float("nan"), device=dev) for _ in range(3))
        dl = torch.full_like(lg, float("nan"))
        args = [tb.desc(), ptr(lg), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, npos, LOSS["hybrid"], 0.01, 0.5, ptr(go),
                ptr(vb), ptr(ce), ptr(loss), ptr(dl), None, None, 0]
        if static:
            check(lib.d3pm_hybrid_loss_static(*args, stream_ptr()))
        else:
            wsb = lib.d3pm_hybrid_loss_workspace(B, npos, K)
            ws = torch.empty(max(wsb // 4, 1), device=dev)
            check(lib.d3pm_hybrid_loss(*args, ptr(ws), wsb, stream_ptr()))
        torch.cuda.synchronize()
        outs.append([o.cpu().numpy() for o in (vb, ce, loss, dl)])
    for a, b in zip(*outs):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("kind,bits,T", [("uniform", 5, 50), ("gaussian", 4, 20)])
def test_static_scratch_matches_workspace_variant(kind, bits, T):
    import d3pm_amd
    dev = torch.device("cuda:0")
    betas = d3pm_amd.generate_betas(0.0, 0.0, T, "jsd") if kind == "uniform" else \
        d3pm_amd.generate_betas(1e-3, 0.2, T, "linear")
    d = d3pm_amd.DiscreteDiffusion(betas=betas, transition_mat_type=kind, num_bits=bits, transition_bands=None,
                                   model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                   hybrid_coeff=0.01, device=dev)
    assert d.tables().structured == (kind == "uniform")
    _both(d, 8, 300, seed=1)          # split over chunks: partial sums in the scratch
    _both(d, 3, 50, seed=2)           # smaller: stale partials of the first call lie behind this call's
    _both(d, 200, 40, seed=3)         # one workgroup per batch element when structured


def test_static_scratch_rejects_oversize():
    """A dense problem whose launch needs more than 16 384 workgroups is refused with a clear error, nothing runs."""
    import d3pm_amd
    from d3pm_amd._lib import lib, ptr, stream_ptr, LOSS, EINVAL
    dev = torch.device("cuda:0")
    d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(1e-3, 0.2, 20, "linear"),
                                   transition_mat_type="gaussian", num_bits=4, transition_bands=None,
                                   model_prediction="x_start", model_output="logits", loss_type="hybrid",
                                   hybrid_coeff=0.01, device=dev)
    B, npos, K = 20000, 1, 16
    x0 = torch.zeros(B, npos, dtype=torch.int64, device=dev)
    t = torch.ones(B, dtype=torch.int32, device=dev)
    lg = torch.zeros(B, npos, K, device=dev)
    loss = torch.full((B,), -7.0, device=dev)
    rc = lib.d3pm_hybrid_loss_static(d.tables().desc(), ptr(lg), K, ptr(x0), 8, ptr(x0), 8, ptr(t), 4, B, npos,
                                     LOSS["hybrid"], 0.01, 1.0, None, None, None, ptr(loss), None, None, None, 0,
                                     stream_ptr())
    torch.cuda.synchronize()
    assert rc == EINVAL and b"static scratch" in lib.d3pm_last_error()
    assert (loss == -7.0).all()


def test_static_scratch_under_graph_capture():
    """Warm (the scratch array's address is resolved at set-up), then capture a HIP graph that calls the entry point
    with a chunked launch (partial sums in the scratch), replay it twice on fresh logits, and compare every output
    with d3pm_hybrid_loss on the same inputs: the captured call allocated nothing and reads no stale state."""
    import d3pm_amd
    from d3pm_amd._lib import lib, check, ptr, stream_ptr, LOSS
    dev = torch.device("cuda:0")
    d = d3pm_amd.DiscreteDiffusion(betas=d3pm_amd.generate_betas(0.0, 0.0, 50, "jsd"), transition_mat_type="uniform",
                                   num_bits=5, transition_bands=None, model_prediction="x_start", model_output="logits",
                                   loss_type="hybrid", hybrid_coeff=0.01, device=dev)
    tb = d.tables()                          # d3pm_digest_f32 has resolved the array on this device
    B, npos, K = 8, 300, 32
    g = torch.Generator().manual_seed(4)
    x0 = torch.randint(0, K, (B, npos), generator=g).to(dev)
    t = torch.randint(1, 50, (B,), generator=g, dtype=torch.int32).to(dev)
    xt = d.q_sample(x0, t, torch.rand(B, npos, K, generator=g).to(dev))
    lg = torch.randn(B, npos, K, generator=g).to(dev)
    loss, dl = torch.empty(B, device=dev), torch.empty_like(lg)

    def call(stream):
        check(lib.d3pm_hybrid_loss_static(tb.desc(), ptr(lg), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, npos,
                                          LOSS["hybrid"], 0.01, 0.125, None, None, None, ptr(loss), ptr(dl), None, None,
                                          0, stream))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(stream_ptr())
    for seed in (5, 6):
        lg.copy_(torch.randn(B, npos, K, generator=torch.Generator().manual_seed(seed)).to(dev))
        loss.fill_(float("nan")); dl.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        want_loss, want_dl = torch.empty(B, device=dev), torch.empty_like(lg)
        wsb = lib.d3pm_hybrid_loss_workspace(B, npos, K)
        ws = torch.empty(max(wsb // 4, 1), device=dev)
        check(lib.d3pm_hybrid_loss(tb.desc(), ptr(lg), K, ptr(x0), 8, ptr(xt), 8, ptr(t), 4, B, npos, LOSS["hybrid"], 0.01,
                                   0.125, None, None, None, ptr(want_loss), ptr(want_dl), None, None, 0, ptr(ws), wsb,
                                   stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(loss, want_loss) and torch.equal(dl, want_dl)
