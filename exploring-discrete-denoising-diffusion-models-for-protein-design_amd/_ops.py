"""``torch.ops.d3pm_mi355x``: the PyTorch-ROCm extension form of the boundary
(``csrc/torch_ops.cpp``, a thin shim over the C ABI of libd3pm_hip.so).

Every call the facade makes goes through these ops (they make the tensors' device
current, check shapes and allocate outputs); tests, the benchmark's kernel
timings and foreign callers use the C ABI directly through ``_lib`` (ctypes).  Both libraries are built in-tree by
``__graft_entry__.build()``; importing this module without them raises."""
from __future__ import annotations

import os

import torch

from . import _lib  # noqa: F401  (libd3pm_hip.so must be loaded first: the shim links against it)

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_PATH = os.environ.get("D3PM_OPS_LIB", os.path.join(_HERE, "libd3pm_torch_ops.so"))
if not os.path.exists(OPS_PATH):
    raise ImportError(f"{OPS_PATH} is missing: build the extension first "
                      "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback")
torch.ops.load_library(OPS_PATH)
ops = torch.ops.d3pm_mi355x

OP_NAMES = ("q_probs", "q_sample", "q_sample_draw_t", "q_posterior_logits", "q_posterior_logits_backward", "hybrid_loss",
            "hybrid_loss_step", "hybrid_loss_then_q_sample", "hybrid_loss_static_", "scale_rows_", "p_sample", "prior_bpd",
            "logistic_logits", "logistic_logits_backward", "build_tables", "protein_uniform_tables", "cumprod",
            "digest_f32", "digest_f64", "table_to_f32", "table_split_f16x2", "table_fingerprints", "lvb_digest", "lvb_loss", "lvb_hybrid_loss", "lvb_reverse_step", "corrupt")


def i64(x: int) -> int:
    """A 64-bit Philox seed / offset as the signed integer an op schema carries."""
    x = int(x) & 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x
