"""Importable alias for the hyphenated package directory `raytracing-in-a-weekend_amd/`."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("raytracing-in-a-weekend_amd")
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
pkg = _pkg
