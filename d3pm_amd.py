"""Import alias: ``import d3pm_amd`` loads the package directory
``exploring-discrete-denoising-diffusion-models-for-protein-design_amd``
(whose name is not a Python identifier) and re-exports it, sub-modules
included, so ``from d3pm_amd.d3pm import utils`` names the same module objects."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_LONG = "exploring-discrete-denoising-diffusion-models-for-protein-design_amd"
_pkg = importlib.import_module(_LONG)
for _name, _mod in list(sys.modules.items()):
    if _name == _LONG or _name.startswith(_LONG + "."):
        sys.modules[__name__ + _name[len(_LONG):]] = _mod
