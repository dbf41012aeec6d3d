"""Importable alias for the package directory ``face-landmark-detector_amd``.

The mandated directory name contains hyphens, which ``import`` statements cannot spell.
``import flm_amd`` (and ``import flm_amd.x.y``) resolve to the very same module objects as
``face-landmark-detector_amd`` (``.x.y``): a meta-path finder aliases names instead of loading
the files a second time, so there is one copy of every module (one library handle, one set of
exception classes).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

_ALIAS = __name__
_REAL = "face-landmark-detector_amd"
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

_real = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _real
