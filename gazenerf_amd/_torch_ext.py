"""Loader of the PyTorch-ROCm C++ binding (gazenerf_amd/_gnr_torch.so, built from csrc/gnr_torch.cpp by
``python -m gazenerf_amd.build``).  The binding validates tensors with TORCH_CHECK, sets the device guard, takes the
current HIP stream in C++ and calls the C ABI of libgnr.so; the ctypes binding (_lib.py) is the second consumer of the
same ABI.  ``GNR_BINDING=torch_ext|ctypes`` forces one; by default the C++ binding is used when it has been built."""
from __future__ import annotations

import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_gnr_torch.so")
_mod = None
_tried = False


def load(required: bool = False):
    """The extension module, or None when it has not been built (``required``: raise instead)."""
    global _mod, _tried
    if _mod is None and not _tried:
        _tried = True
        if os.path.exists(PATH):
            import torch  # noqa: F401  (libtorch must be loaded before the extension)
            from . import _lib
            _lib.load()                                   # fails loudly if libgnr.so itself is missing
            spec = importlib.util.spec_from_file_location("_gnr_torch", PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != _lib.ABI_VERSION:
                raise RuntimeError("_gnr_torch.so was built against ABI %d, libgnr.so is ABI %d; rebuild" % (
                    mod.abi_version(), _lib.ABI_VERSION))
            from ._srchash import source_hash
            if source_hash() is not None and mod.source_hash() != source_hash():
                raise RuntimeError("_gnr_torch.so was built from other sources than this tree (%s vs %s); rebuild with "
                                   "`python -m gazenerf_amd.build`" % (mod.source_hash(), source_hash()))
            _mod = mod
    if _mod is None and required:
        raise RuntimeError("gazenerf_amd: %s is missing. Build it with `python -m gazenerf_amd.build`." % PATH)
    return _mod


def active():
    """The module to use for the render op under the current GNR_BINDING setting (None == ctypes)."""
    want = os.environ.get("GNR_BINDING", "auto")
    if want == "ctypes":
        return None
    if want == "torch_ext":
        return load(required=True)
    if want != "auto":
        raise ValueError("GNR_BINDING must be auto, torch_ext or ctypes")
    return load()
