"""
abdpymc_amd -- MI355X-native joint log-probability (+ gradient) of the abdpymc antibody-dynamics model.

Drop-in for ONE path of davipatti/abdpymc: the two callables PyMC compiles out of
``abdpymc.model(data, splits, ignore_pcrpos)`` (reference abdpymc/abd.py:396-442, call site abd.py:922).
Host code is Python over a C ABI (``include/abd_hip.h``); the arithmetic is hand-written HIP for gfx950.
"""
__version__ = "0.1.0"

from . import synthetic  # noqa: F401  (pure NumPy)

# the reference exports these at package level (abdpymc/__init__.py:3-18): abdpymc.model(...), abdpymc.TiterData.
# Importing them does not touch HIP: the library is loaded when the first model is built.
from .data import AntigenTiterData, TiterData, check_splits  # noqa: E402,F401
from .model import AbdModel, model  # noqa: E402,F401  (after this, ``abdpymc_amd.model`` is the FUNCTION, as in the reference)
from .sampler import sample  # noqa: E402,F401

__all__ = ["synthetic", "TiterData", "AntigenTiterData", "check_splits", "model", "AbdModel", "sample"]
