"""MI355X-native static-Huffman compression of binned multi-channel MUA spike counts.

A from-scratch gfx950 implementation of the one data-parallel hot path of
zhengzhang96/Hardware-efficient-MUA-compression (per-channel calibration histogram,
approximate-sort mapper, static-Huffman encoder selection, bit totals) plus the bit-packer,
container and decoder the reference only models.  Kernels live in ``csrc/`` behind the C ABI
of ``include/muahuff.h`` (``libmuahuff.so``); this package is the Python host side.
There is no CPU fallback: without the built library and a GPU, device operations raise.
"""
import importlib.abc as _abc
import importlib.util as _util
import sys as _sys


class _Alias(_abc.MetaPathFinder, _abc.Loader):
    """``muahuff`` and ``muahuff.<sub>`` are THIS package and its submodules (the directory name,
    fixed by the project layout, is not a Python identifier).  Without this finder
    ``from muahuff import codec`` would load a second copy of every submodule under the alias
    name -- two library handles, two MuaHuffError classes."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname == "muahuff" or fullname.startswith("muahuff."):
            return _util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        import importlib
        return importlib.import_module(__name__ + spec.name[len("muahuff"):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _Alias) for f in _sys.meta_path):
    _sys.meta_path.insert(0, _Alias())
_sys.modules["muahuff"] = _sys.modules[__name__]

from . import _lib, sclv  # noqa: E402,F401
from ._lib import (CHUNK, MODE_APPROX, MODE_NOSORT, WIN_AFTER_CAL, WIN_FULL,  # noqa: F401
                   WIN_REF_HALF, WIN_REF_HALF_TRUNC, WIN_REV2_SEGMENTS, MuaHuffError, device_info)

__version__ = "0.1.0"


def __getattr__(name):  # torch-dependent modules load lazily
    import importlib
    if name in ("bit_rates", "compress", "decompress"):
        return getattr(importlib.import_module(".api", __name__), name)
    if name in ("codec", "container", "container_io", "synth", "api", "sweep", "stream", "analysis", "functions_1", "drivers", "dist"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
