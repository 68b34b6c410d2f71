import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    conv = lambda a: a.item() if a.ndim == 0 else (a if a.dtype.kind in "US" else torch.from_numpy(a))
    return {k: conv(z[k]) for k in z.files}


def golden_problem(g, device="cpu"):
    keys = ("xy", "R", "T", "Kinv", "shape_code", "appea_code", "gaze")
    return {k: g["in_" + k].to(device) for k in keys}


@pytest.fixture(scope="session")
def lib_built():
    """Build (or reuse) libgnr.so; hipcc cross-compiles without a GPU."""
    from gazenerf_amd import build
    return build.build(verbose=False)
