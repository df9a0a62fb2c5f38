"""Import alias: ``import mi355seg`` == the package in
``general-medical-image-segmentation-cnn-framework_amd/`` (whose directory name is not a
valid Python identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("general-medical-image-segmentation-cnn-framework_amd")
sys.modules[__name__] = _pkg
