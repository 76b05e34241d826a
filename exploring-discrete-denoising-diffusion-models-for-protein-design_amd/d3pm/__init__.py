"""Image-style D3PM process: the call surface of the reference's ``d3pm/``
sub-project (``diffusion_discrete.py``, ``utils.py``) on HIP kernels."""
from .diffusion_discrete import DiscreteDiffusion, generate_betas  # noqa: F401
from . import utils  # noqa: F401
