"""Import alias: ``import muahuff`` == the package ``hardware-efficient-mua-compression_amd``
(whose directory name, fixed by the project layout, is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("hardware-efficient-mua-compression_amd")
