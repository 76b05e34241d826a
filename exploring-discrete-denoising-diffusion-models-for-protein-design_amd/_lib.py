"""ctypes binding of libd3pm_hip.so (the C ABI declared in include/d3pm_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no fallback: if the shared object is missing or a symbol cannot be
resolved, importing this module raises, and every operator of the package
depends on it.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D3PM_LIB", os.path.join(_HERE, "libd3pm_hip.so"))

OK, EINVAL, EUNSUPPORTED, EHIP = 0, 1, 2, 3
KIND = {"uniform": 0, "gaussian": 1, "absorbing": 2}
NOISE_UNIFORM, NOISE_GUMBEL, NOISE_PHILOX, NOISE_EXPONENTIAL = 0, 1, 2, 3
NOISE = {"uniform": NOISE_UNIFORM, "gumbel": NOISE_GUMBEL, "philox": NOISE_PHILOX}
LOSS = {"kl": 0, "cross_entropy_x_start": 1, "hybrid": 2}


class TablesF32(C.Structure):
    _fields_ = [("K", C.c_int), ("T", C.c_int), ("structured", C.c_int), ("eps", C.c_float),
                ("qbar_digest", C.c_void_p), ("qone_digest", C.c_void_p), ("qbar", C.c_void_p),
                ("qoneT", C.c_void_p), ("qbarT", C.c_void_p), ("qbar_f16x2", C.c_void_p),
                ("qbarT_f16x2", C.c_void_p)]


class TablesF64(C.Structure):
    _fields_ = [("K", C.c_int), ("T", C.c_int), ("qbar_digest", C.c_void_p), ("qone_digest", C.c_void_p),
                ("lvb_digest", C.c_void_p)]


_P, _I, _L, _F, _D, _U64, _SZ = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_uint64, C.c_size_t
_T32, _T64 = C.POINTER(TablesF32), C.POINTER(TablesF64)

# name -> (restype, argtypes); must list every function of include/d3pm_hip.h
SIGNATURES = {
    "d3pm_last_error": (C.c_char_p, []),
    "d3pm_abi_version": (_I, []),
    "d3pm_build_tables_f64": (_I, [_I, _I, _I, _I, _P, _P, _P, _P]),
    "d3pm_protein_uniform_tables_f64": (_I, [_I, _I, _P, _P, _P, _P, _P]),
    "d3pm_cumprod_f64": (_I, [_I, _I, _P, _L, _L, _L, _P, _P]),
    "d3pm_digest_f32": (_I, [_P, _L, _L, _L, _I, _I, _F, _P, _P, _P]),
    "d3pm_digest_f64": (_I, [_P, _L, _L, _L, _I, _I, _D, _P, _P, _P]),
    "d3pm_table_fingerprint": (_I, [_P, _L, _L, _L, _I, _I, _P, _P]),
    "d3pm_table_to_f32": (_I, [_P, _L, _L, _L, _I, _I, _P, _P]),
    "d3pm_table_split_f16x2": (_I, [_P, _I, _I, _P, _P]),
    "d3pm_q_probs": (_I, [_T32, _P, _I, _P, _I, _I, _L, _P, _P]),
    "d3pm_q_sample": (_I, [_T32, _P, _I, _P, _I, _I, _L, _P, _I, _U64, _U64, _P, _P, _P]),
    "d3pm_q_posterior_logits": (_I, [_T32, _P, _I, _L, _I, _P, _I, _P, _I, _I, _L, _P, _P]),
    "d3pm_q_posterior_logits_backward": (_I, [_T32, _P, _L, _P, _I, _I, _L, _P, _P, _P]),
    "d3pm_hybrid_loss_workspace": (_SZ, [_I, _L, _I]),
    "d3pm_q_sample_draw_t": (_I, [_T32, _P, _I, _I, _L, _U64, _U64, _P, _P, _P, _P, _P]),
    "d3pm_hybrid_loss": (_I, [_T32, _P, _L, _P, _I, _P, _I, _P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _U64,
                              _P, _SZ, _P]),
    "d3pm_hybrid_loss_then_q_sample": (_I, [_T32, _P, _L, _P, _I, _P, _I, _P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P,
                                            _SZ, _P, _I, _U64, _U64, _P, _P, _P, _P]),
    "d3pm_hybrid_loss_static": (_I, [_T32, _P, _L, _P, _I, _P, _I, _P, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P,
                                     _U64, _P]),
    "d3pm_scale_rows": (_I, [_P, _P, _F, _I, _L, _P]),
    "d3pm_p_sample": (_I, [_T32, _P, _L, _P, _I, _P, _I, _I, _L, _P, _I, _U64, _U64, _P, _P, _P, _P]),
    "d3pm_prior_bpd": (_I, [_T32, _P, _I, _I, _L, _I, _P, _P, _SZ, _P]),
    "d3pm_logistic_logits": (_I, [_P, _P, _P, _L, _I, _F, _P, _P]),
    "d3pm_logistic_logits_backward": (_I, [_P, _P, _P, _P, _L, _I, _F, _P, _P, _P]),
    "d3pm_corrupt": (_I, [_T64, _P, _P, _I, _L, _P, _U64, _U64, _P, _P, _P]),
    "d3pm_lvb_digest_f32_size": (_SZ, [_I]),
    "d3pm_lvb_digest_f32": (_I, [_T64, _P, _P]),
    "d3pm_lvb_loss_workspace": (_SZ, [_I, _L, _I]),
    "d3pm_lvb_loss": (_I, [_T64, _P, _L, _P, _P, _P, _P, _I, _L, _I, _I, _D, _P, _P, _P, _SZ, _P]),
    "d3pm_lvb_hybrid_loss": (_I, [_T64, _P, _L, _P, _P, _P, _P, _I, _L, _I, _I, _D, _D, _P, _P, _P, _P, _SZ, _P]),
    "d3pm_lvb_reverse_step": (_I, [_T64, _P, _L, _P, _P, _I, _L, _P, _U64, _U64, _P, _D, _L, _P, _P]),
}


ABI_VERSION = 10       # d3pm_abi_version() of the library these signatures describe


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    have = lib.d3pm_abi_version()
    if have != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {have}, this binding was written for {ABI_VERSION}: rebuild it "
                          "(python -c 'import __graft_entry__ as g; g.build()')")
    return lib


lib = _load()


def check(rc: int):
    """Map a status code to the exception the reference would have raised."""
    if rc == OK:
        return
    msg = (lib.d3pm_last_error() or b"").decode()
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == EUNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"libd3pm_hip: {msg} (code {rc})")


def stream_ptr() -> int:
    """The current torch HIP stream: every kernel is enqueued there."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libd3pm_hip operators need tensors on an MI355X (device='cuda'); "
                               "there is no CPU path in this package")


def tok_bytes(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return 8
    if t.dtype == torch.int32:
        return 4
    raise TypeError(f"token/timestep tensors must be int32 or int64, got {t.dtype}")
