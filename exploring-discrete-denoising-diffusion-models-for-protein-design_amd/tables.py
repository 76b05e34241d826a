"""Device-resident transition tables and the digests the kernels read.

``ImageTables`` serves the image-style process (float32 semantics of
``DiscreteDiffusion._at`` / ``_at_onehot``); ``ProteinTables`` serves the
protein-style kernels (float64).  Both are built once per (Q_t, Qbar) pair and
cached: the reference re-casts (image style) or re-uploads (protein style) the
whole tables on every call.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import ptr
from ._ops import ops


def _strides3(t: torch.Tensor) -> Tuple[int, int, int]:
    assert t.dim() == 3
    return t.stride(0), t.stride(1), t.stride(2)


class ImageTables:
    """float32 view of (one-step Q_t, cumulative Qbar_t) for d3pm-style kernels.

    ``q_onestep`` is indexed [t, from, to]; pass ``transposed_onestep=True`` if
    the tensor is the reference's ``transpose_q_onestep_mats`` ([t, to, from]).
    Column-structured tables (uniform, absorbing) get the O(K) digest path;
    anything else (Gaussian, banded) the dense path with float32 copies.

    ``dense_matmul`` picks how the K x K contraction of ``_at_onehot`` runs for
    dense tables with K = 64, 128 or 256 (at 64 and 128 in the training loss
    only): ``"f16x2"`` (default) emulates float32 on the
    float16 matrix instructions -- every operand, scaled into float16's range,
    split into two float16 pieces, three products per term, relative error below
    2^-22 per product (absolute 2^-39 for entries below 2^-16) -- from split
    copies of the cumulative table built here (4 bytes per entry each way);
    ``"f32"`` uses the float32 matrix instructions at K = 256 (an in-order
    float32 fma chain) and the vector ALUs otherwise.  Other alphabets contract
    on the vector ALUs in float32 either way.
    """

    def __init__(self, q_mats: torch.Tensor, q_onestep: torch.Tensor, eps: float = 1e-6,
                 transposed_onestep: bool = False, force_dense: bool = False, dense_matmul: str = "f16x2"):
        if dense_matmul not in ("f16x2", "f32"):
            raise ValueError(f"dense_matmul must be 'f16x2' or 'f32', got {dense_matmul!r}")
        _lib.require_device(q_mats, q_onestep)
        if q_mats.dtype != torch.float64 or q_onestep.dtype != torch.float64:
            q_mats, q_onestep = q_mats.double(), q_onestep.double()
        assert q_mats.dim() == 3 and q_mats.shape == q_onestep.shape and q_mats.shape[1] == q_mats.shape[2]
        self.T, self.K = int(q_mats.shape[0]), int(q_mats.shape[1])
        self.eps = float(eps)
        self.device = q_mats.device
        if transposed_onestep:
            q_onestep = q_onestep.permute(0, 2, 1)       # back to [t, from, to], no copy
        self._keep = (q_mats, q_onestep)
        T, K = self.T, self.K
        self.qbar_digest, v0 = ops.digest_f32(q_mats, self.eps)
        self.qone_digest, v1 = ops.digest_f32(q_onestep, self.eps)
        self.violations = (int(v0.item()), int(v1.item()))               # one sync each, at set-up
        self.structured = (self.violations == (0, 0)) and not force_dense
        self.qbar = self.qoneT = self.qbarT = self.qbar_f16x2 = self.qbarT_f16x2 = None
        if not self.structured:
            self.qbar = ops.table_to_f32(q_mats)
            self.qbarT = ops.table_to_f32(q_mats.permute(0, 2, 1))
            self.qoneT = ops.table_to_f32(q_onestep.permute(0, 2, 1))
            # the float16 pieces carry 2^14 x the entry: tables of probabilities only (a caller's own q_mats with
            # entries above 1 -- not a transition table -- stay on the float32 instructions; one host read at set-up)
            if K in (64, 128, 256) and dense_matmul == "f16x2" and float(self.qbar.abs().max()) <= 1.0 + 1e-6:
                self.qbar_f16x2 = ops.table_split_f16x2(self.qbar)
                self.qbarT_f16x2 = ops.table_split_f16x2(self.qbarT)
        self.dense_matmul = dense_matmul if self.qbar_f16x2 is not None else "f32"
        self._desc = _lib.TablesF32(K, T, int(self.structured), self.eps, ptr(self.qbar_digest),
                                    ptr(self.qone_digest), ptr(self.qbar), ptr(self.qoneT), ptr(self.qbarT),
                                    ptr(self.qbar_f16x2), ptr(self.qbarT_f16x2))

    def desc(self):
        return C.byref(self._desc)

    def op_args(self):
        """Leading arguments of the torch.ops.d3pm_mi355x image-style operators."""
        a = getattr(self, "_op_args", None)
        if a is None:
            e = torch.empty(0, dtype=torch.float32, device=self.device)
            tabs = [self.qbar_digest, self.qone_digest, e if self.qbar is None else self.qbar,
                    e if self.qoneT is None else self.qoneT, e if self.qbarT is None else self.qbarT]
            if self.qbar_f16x2 is not None:
                tabs += [self.qbar_f16x2, self.qbarT_f16x2]
            a = self._op_args = (tabs, self.K, self.T, bool(self.structured), self.eps)
        return a


class ProteinTables:
    """float64 column-structured digests of (Q_t, Qbar_t) for the protein-style
    kernels.  Raises NotImplementedError for tables without that structure."""

    RTOL = 1e-9

    def __init__(self, Q: torch.Tensor, Q_bar: torch.Tensor):
        _lib.require_device(Q, Q_bar)
        assert Q.dim() == 3 and Q.shape == Q_bar.shape and Q.shape[1] == Q.shape[2]
        if Q.dtype != torch.float64 or Q_bar.dtype != torch.float64:
            Q, Q_bar = Q.double(), Q_bar.double()
        self.T, self.K = int(Q.shape[0]), int(Q.shape[1])
        self.device = Q.device
        self._keep = (Q, Q_bar)
        T, K = self.T, self.K
        self.qbar_digest, v0 = ops.digest_f64(Q_bar, self.RTOL)
        self.qone_digest, v1 = ops.digest_f64(Q, self.RTOL)
        self.violations = (int(v0.item()), int(v1.item()))
        if self.violations != (0, 0):
            raise NotImplementedError(
                "protein-style kernels need column-structured transition tables (uniform schedule); "
                f"{self.violations} entries break the structure")
        # float32 constants of the training form of the LVB loss (alphabets of up to 32 classes)
        self.lvb_digest = ops.lvb_digest(self.qbar_digest, self.qone_digest, K, T) if K <= 32 else \
            torch.empty(0, dtype=torch.float32, device=self.device)
        self._desc = _lib.TablesF64(K, T, ptr(self.qbar_digest), ptr(self.qone_digest),
                                    ptr(self.lvb_digest) if self.lvb_digest.numel() else None)

    def desc(self):
        return C.byref(self._desc)

    def op_args(self):
        """Leading arguments of the torch.ops.d3pm_mi355x protein-style operators."""
        return (self.qbar_digest, self.qone_digest, self.lvb_digest, self.K, self.T)


# Two keys lead to a ProteinTables: the identity of the caller's tensors (storage pointer + version: no device work
# at all when the same tensors come back, the collaters' own `self.Q` for instance) and, when that misses, the
# tables' content (a 64-bit fingerprint per table: one short launch each and one host read).  The reference's
# training loop uploads Q and Q_bar anew on every step (Protein Generation Notebook.ipynb cell-12:
# `Q = Q.to(device)`), so the loss sees fresh pointers with known content step after step; the content key turns
# that into a hit instead of two digest builds and their structure check.
_by_identity: Dict[tuple, ProteinTables] = {}
_by_content: Dict[tuple, ProteinTables] = {}
_MAX_ENTRIES = 8
builds = 0          # ProteinTables constructed by protein_tables() so far (read by the tests)


def _host_fingerprint(x: torch.Tensor) -> int:
    """The fingerprint of d3pm_table_fingerprint evaluated on the host (numpy, modulo 2^64): a table has one content
    key wherever it lives."""
    import numpy as np
    bits = x.detach().cpu().to(torch.float64).contiguous().numpy().reshape(-1).view(np.uint64)
    with np.errstate(over="ignore"):
        z = np.arange(bits.size, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        f = int((bits * (z | np.uint64(1))).sum(dtype=np.uint64))
    return f - (1 << 64) if f >= (1 << 63) else f          # as the int64 the device tensor holds


def _content_key(Q: torch.Tensor, Q_bar: torch.Tensor, dev: torch.device) -> tuple:
    if Q.is_cuda and Q_bar.is_cuda and Q.dtype == Q_bar.dtype == torch.float64 and Q.device == Q_bar.device:
        fq, fb = ops.table_fingerprints([Q, Q_bar]).tolist()       # two short launches, one host read
    else:
        fq, fb = _host_fingerprint(Q), _host_fingerprint(Q_bar)     # hashed where they are, before any upload
    return (fq, fb, tuple(Q.shape), str(dev))


def protein_tables(Q: torch.Tensor, Q_bar: torch.Tensor, device=None) -> ProteinTables:
    """Cached ProteinTables for the (Q, Q_bar) pair a caller passes on every
    call (losses.py:48, generate.py:8).  CPU tensors are uploaded once."""
    global builds
    dev = torch.device(device) if device is not None else Q.device
    ident = (Q.data_ptr(), Q._version, Q_bar.data_ptr(), Q_bar._version, tuple(Q.shape), str(dev))
    hit = _by_identity.get(ident)
    if hit is not None:
        return hit
    if dev.type != "cuda":
        raise RuntimeError("protein-style kernels need a HIP device; there is no CPU path in this package")
    assert Q.dim() == 3 and Q.shape == Q_bar.shape and Q.shape[1] == Q.shape[2], (Q.shape, Q_bar.shape)
    content = _content_key(Q, Q_bar, dev)
    tb = _by_content.get(content)
    if tb is not None:
        # a fresh copy of known tables: nothing is kept of it (its pointer may be recycled for other content)
        return tb
    tb = ProteinTables(Q.to(dev), Q_bar.to(dev))
    builds += 1
    tb._src = (Q, Q_bar)           # keep the identity key's storage alive so its data_ptr cannot be recycled
    for cache in (_by_identity, _by_content):
        while len(cache) >= _MAX_ENTRIES:
            cache.pop(next(iter(cache)))           # oldest entry first
    _by_identity[ident] = tb
    _by_content[content] = tb
    return tb
