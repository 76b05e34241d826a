"""``torch.ops.d3pm_mi355x``: the PyTorch-ROCm extension form of the boundary
(``csrc/torch_ops.cpp``, a thin shim over the C ABI of libd3pm_hip.so).

The facade's reference-shaped operators go through these ops; set-up (table
building), the extended two-launch training step and the micro-benchmarks call
the C ABI directly through ``_lib`` (ctypes).  Both libraries are built in-tree by
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

OP_NAMES = ("q_probs", "q_sample", "q_posterior_logits", "q_posterior_logits_backward", "hybrid_loss", "scale_rows_",
            "p_sample", "prior_bpd", "lvb_loss", "lvb_reverse_step", "corrupt")


def i64(x: int) -> int:
    """A 64-bit Philox seed / offset as the signed integer an op schema carries."""
    x = int(x) & 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x
