"""CPU oracle for the D3PM categorical-diffusion core.

TEST INFRASTRUCTURE ONLY.  Nothing under this package is part of the shipped
product path: it is imported by ``tests/``, by ``__graft_entry__.smoke()`` and
by the ``cpu_baseline`` leg of ``bench.py``, always as the checker (or as the
timed CPU baseline), never as the implementation of an operator.

The modules restate, with eager PyTorch-CPU / numpy ops in the same order and
dtypes, the algorithms of the reference files they cite:

* ``image_style``   -> d3pm/diffusion_discrete.py + d3pm/utils.py
* ``protein_style`` -> protein_generation/{utils,losses,generate}.py

Pinning: every function here is checked against golden vectors produced by
importing the reference itself (``tools/make_golden.py`` -> ``tests/golden``).
"""
