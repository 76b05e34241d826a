"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares (no compute calls), and the host logic of the facade."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    ge.build()
    import d3pm_amd
    return d3pm_amd


def header_functions():
    src = open(os.path.join(ROOT, "include", "d3pm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3pm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/d3pm_hip.h but not exported"
    # and the binding lists exactly the header's functions
    assert sorted(pkg._lib.SIGNATURES) == names
    assert pkg._lib.lib.d3pm_abi_version() == pkg._lib.ABI_VERSION == 10


def test_torch_ops_registered(pkg):
    """The PyTorch-extension form of the boundary: libd3pm_torch_ops.so loads and registers every operator of
    torch.ops.d3pm_mi355x (no compute calls here), and the operators refuse CPU tensors."""
    for name in pkg._ops.OP_NAMES:
        assert hasattr(torch.ops.d3pm_mi355x, name), name
    schema = torch.ops.d3pm_mi355x.hybrid_loss.default._schema
    assert [a.name for a in schema.arguments][:5] == ["tables", "K", "T", "structured", "eps"]
    assert len(schema.returns) == 4
    x = torch.zeros(2, 3, dtype=torch.int64)
    e = torch.empty(0)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.d3pm_mi355x.q_probs([e, e, e, e, e], 4, 2, True, 1e-6, x, torch.zeros(2, dtype=torch.int64))


def test_struct_layout_matches_header(pkg):
    # d3pm_tables_f32: 3 ints + float + 7 pointers; d3pm_tables_f64: 2 ints + 3 pointers
    assert ctypes.sizeof(pkg._lib.TablesF32) == 16 + 7 * 8
    assert ctypes.sizeof(pkg._lib.TablesF64) == 8 + 3 * 8


def test_generate_betas_matches_golden(pkg):
    g = golden("image_misc")
    np.testing.assert_array_equal(pkg.generate_betas(1e-4, 0.02, 1000, "linear").numpy(), g["betas_linear"])
    np.testing.assert_array_equal(pkg.generate_betas(0, 0, 500, "cosine").numpy(), g["betas_cosine"])
    np.testing.assert_array_equal(pkg.generate_betas(0, 0, 500, "jsd").numpy(), g["betas_jsd"])
    with pytest.raises(NotImplementedError):
        pkg.generate_betas(0, 1, 10, "quadratic")


def test_protein_schedule_and_tokenizer(pkg):
    from d3pm_amd.protein_generation import utils as pu, constants as c
    g = golden("protein_tables_k22")
    np.testing.assert_array_equal(pu._beta_schedule(500, schedule="sohl-dickstein").numpy(), g["betas_sohl"])
    np.testing.assert_array_equal(pu._beta_schedule(500, schedule="linear").numpy(), g["betas_linear"])
    np.testing.assert_array_equal(pu._beta_schedule(500, schedule="cosine").numpy(), g["betas_cosine"])
    with pytest.raises(NotImplementedError):
        pu._beta_schedule(10, schedule="nope")
    assert c.PROTEIN_ALPHABET == "ACDEFGHIKLMNPQRSTVWYX-#!" and c.MSA_AAS == "ACDEFGHIKLMNPQRSTVWYX-"
    tok = pu.Tokenizer()
    assert tok.K == 22
    s = "ACD-XY"
    ids = tok.tokenize(s)
    assert ids.tolist() == [0, 1, 2, 21, 20, 19]
    assert tok.untokenize(ids) == s and tok.untokenize(torch.tensor(ids)) == s
    oh = tok.one_hot(torch.tensor(ids))
    assert oh.dtype == torch.float64 and oh.shape == (6, 22)
    lf = pkg.D3PMLVBLoss(tmax=500, tokenizer=tok, evaluation=True)
    assert isinstance(lf, torch.nn.KLDivLoss) and lf.K == 22 and lf.tmax == 500


def test_constructor_errors_without_gpu(pkg):
    mk = lambda **kw: pkg.DiscreteDiffusion(**{**dict(
        betas=pkg.generate_betas(0, 0, 10, "jsd"), transition_mat_type="uniform", num_bits=3, transition_bands=None,
        model_prediction="x_start", model_output="logits", loss_type="hybrid", hybrid_coeff=0.001,
        device="cuda"), **kw})
    with pytest.raises(ValueError):
        mk(betas=[0.1, 0.2])
    with pytest.raises(ValueError):
        mk(betas=torch.tensor([0.0, 0.5]))
    with pytest.raises(ValueError):
        mk(transition_mat_type="bogus")
    with pytest.raises(RuntimeError):          # no CPU path: fail loudly instead of falling back
        mk(device="cpu")


def test_c_abi_argument_checks_without_gpu(pkg):
    """Argument validation happens before any HIP call, so it can run here."""
    L = pkg._lib
    rc = L.lib.d3pm_build_tables_f64(7, 8, 4, -1, None, None, None, None)
    assert rc == L.EINVAL and b"transition_mat_type" in L.lib.d3pm_last_error()
    with pytest.raises(ValueError):
        L.check(rc)
    tb = L.TablesF32(300, 5, 1, 1e-6, 1, 1, None, None, None)
    rc = L.lib.d3pm_q_probs(ctypes.byref(tb), 1, 8, 1, 8, 1, 1, 1, None)
    assert rc == L.EINVAL
    tb = L.TablesF32(32, 5, 1, 1e-6, 1, 1, None, None, None)
    rc = L.lib.d3pm_hybrid_loss(ctypes.byref(tb), 1, 32, 1, 8, 1, 8, 1, 8, 2, 4, 9, 0.0, 1.0, None, None, None, None,
                                None, None, None, 0, None, 0, None)
    assert rc == L.EUNSUPPORTED
    with pytest.raises(NotImplementedError):
        L.check(rc)
    assert L.lib.d3pm_hybrid_loss_workspace(256, 512, 32) >= 256 * 2 * 4
