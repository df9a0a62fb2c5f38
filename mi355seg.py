"""Import alias: ``import mi355seg`` == the package in
``general-medical-image-segmentation-cnn-framework_amd/`` (whose directory name is not a
valid Python identifier).  Submodules resolve to the SAME module objects under either name
(``mi355seg.functional is <package>.functional``), so there is one library handle, one workspace and one
``Mi355SegError`` class however the package is imported."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ALIAS = __name__
_REAL = "general-medical-image-segmentation-cnn-framework_amd"
_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
