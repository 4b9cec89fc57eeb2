"""
abdpymc_amd -- MI355X-native joint log-probability (+ gradient) of the abdpymc antibody-dynamics model.

Drop-in for ONE path of davipatti/abdpymc: the two callables PyMC compiles out of
``abdpymc.model(data, splits, ignore_pcrpos)`` (reference abdpymc/abd.py:396-442, call site abd.py:922).
Host code is Python over a C ABI (``include/abd_hip.h``); the arithmetic is hand-written HIP for gfx950.
"""
__version__ = "0.1.0"

from . import synthetic  # noqa: F401  (pure NumPy)

__all__ = ["synthetic"]
