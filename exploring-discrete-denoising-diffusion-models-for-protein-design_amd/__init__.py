"""MI355X-native D3PM categorical-diffusion core.

Sub-packages mirror the two reference sub-projects whose hot path this package
replaces: ``d3pm`` (image-style ``DiscreteDiffusion``) and
``protein_generation`` (tokenizer tables, LVB loss, reverse sampling).  All
arithmetic runs in hand-written HIP kernels for gfx950 reached through the C
ABI of ``libd3pm_hip.so`` (``include/d3pm_hip.h``); importing the package
without that library raises.
"""
from . import _lib  # noqa: F401  (loads libd3pm_hip.so or raises)
from . import _ops  # noqa: F401  (registers torch.ops.d3pm_mi355x or raises)
from . import tables  # noqa: F401
from . import graphs  # noqa: F401
from . import d3pm  # noqa: F401
from . import protein_generation  # noqa: F401
from .d3pm import DiscreteDiffusion, generate_betas  # noqa: F401
from .protein_generation import (Tokenizer, D3PMLVBLoss, D3PMLVBLossMSA, generate_d3pm,  # noqa: F401
                                 sample_transition_matrix, cumprod_matrix)

__all__ = ["DiscreteDiffusion", "generate_betas", "Tokenizer", "D3PMLVBLoss", "D3PMLVBLossMSA", "generate_d3pm",
           "sample_transition_matrix", "cumprod_matrix", "d3pm", "protein_generation", "tables"]
