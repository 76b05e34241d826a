"""Protein-style D3PM: the call surface of the reference's
``protein_generation/`` hot path (``utils.py``, ``losses.py``, ``generate.py``,
``constants.py``, the collaters of ``dataset.py``) on HIP kernels."""
from . import constants  # noqa: F401
from .utils import (Tokenizer, cumprod_matrix, _beta_schedule, sample_transition_matrix, CollaterV2,  # noqa: F401
                    custom_mutant_matrix, corrupt_batch)
from .losses import D3PMLVBLoss, D3PMLVBLossMSA, sample_prior, sample_priorMSA, KL_test  # noqa: F401
from .generate import (generate_d3pm, sample_and_save, sample_and_save_sharded, generate_msa_d3pm,  # noqa: F401
                       sample_and_save_msa, reverse_step, shard_range, gather_token_shards)
from .dataset import Collater, CollaterMSA  # noqa: F401
