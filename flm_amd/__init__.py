"""Importable alias for the package directory ``face-landmark-detector_amd``.

The mandated directory name contains hyphens, which ``import`` statements cannot
spell.  ``import flm_amd`` loads that directory through importlib and aliases it,
so ``flm_amd.networks``, ``flm_amd.prediction`` ... are the real modules.
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

_real = importlib.import_module("face-landmark-detector_amd")
sys.modules[__name__] = _real
