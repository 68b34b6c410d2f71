"""Hash of the sources libgnr.so / _gnr_torch.so are built from (csrc/*.hip, *.h, *.cpp and include/gnr.h).

``gazenerf_amd.build`` compiles it into the library (``gnr_build_info()``); ``_lib.load()`` recomputes it from the tree and
refuses a library built from other sources -- the binaries are git-ignored and travel to the GPU box prebuilt, so without
this nothing ties the code that runs to the code that is committed."""
from __future__ import annotations

import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "gnr.h")


def source_files():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    return [os.path.join(CSRC, f) for f in names] + [HEADER]


def source_hash():
    """16 hex digits, or None when the tree carries no sources (an installed copy: nothing to compare with)."""
    if not os.path.isdir(CSRC) or not os.path.exists(HEADER):
        return None
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]
