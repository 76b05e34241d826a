"""Import alias: ``import d3pm_amd`` loads the package directory
``exploring-discrete-denoising-diffusion-models-for-protein-design_amd``
(whose name is not a Python identifier) and re-exports it."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("exploring-discrete-denoising-diffusion-models-for-protein-design_amd")
sys.modules[__name__] = _pkg
