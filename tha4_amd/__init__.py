"""Import alias: ``import tha4_amd`` == the package in ``talking-head-anime-4-demo_amd/``
(that directory name is mandated by the repo layout but is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("talking-head-anime-4-demo_amd")
sys.modules[__name__] = _pkg
