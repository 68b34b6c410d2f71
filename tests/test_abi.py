"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gnr.h declares.
No compute calls here (no GPU in this tier)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib_built):
    lib = ctypes.CDLL(lib_built)
    syms = _declared_symbols()
    assert "gnr_fwd" in syms and "gnr_bwd" in syms and "gnr_resample" in syms
    for s in syms:
        assert hasattr(lib, s), "libgnr.so does not export %s" % s


def test_binding_matches_header(lib_built):
    from gazenerf_amd import _lib
    lib = _lib.load()
    assert lib.gnr_abi_version() == _lib.ABI_VERSION
    assert sorted(_lib.EXPORTS) == _declared_symbols()
    # struct sizes: the library reports its own sizeof() and the binding checked them at load time; spelled out once
    # more here: 8 ints + 2 floats + 9 pointers + 3 ints (+ pad) + 2 pointers, 24 pointers, 8 pointers
    for which, cls in enumerate(_lib.STRUCTS):
        assert lib.gnr_sizeof(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.gnr_sizeof(99) == 0
    assert ctypes.sizeof(_lib.GnrProblem) == 8 * 4 + 2 * 4 + 9 * 8 + 16 + 2 * 8
    assert ctypes.sizeof(_lib.GnrWeights) == 24 * 8
    assert ctypes.sizeof(_lib.GnrOutputs) == 8 * 8
    assert ctypes.sizeof(_lib.GnrMergeProblem) == 16 + 6 * 8        # 3 ints (+pad) + 6 pointers


def test_validation_errors_without_gpu(lib_built):
    """Argument validation happens before any launch, so it is testable on CPU."""
    from gazenerf_amd import _lib
    lib = _lib.load()
    p = _lib.GnrProblem()
    p.batch, p.n_rays, p.n_samples, p.hidden, p.feat_nc = 1, 16, 64, 400, 258
    assert lib.gnr_workspace_bytes(ctypes.byref(p), 2, _lib.WS_FWD) == 0
    assert b"hidden=384" in lib.gnr_last_error()
    p.hidden = 384
    p.xy = p.R = p.T = p.Kinv = 1024          # non-NULL dummies; nothing is dereferenced on the host
    p.shape_dims, p.gaze_dims, p.appea_dims = 179, 2, 127
    p.shape_code = p.gaze = p.appea_code = 1024
    n0 = lib.gnr_workspace_bytes(ctypes.byref(p), 2, _lib.WS_FWD)
    n1 = lib.gnr_workspace_bytes(ctypes.byref(p), 2, _lib.WS_FWD_SAVE)
    assert 0 < n0 < n1
    p.n_samples = 1
    assert lib.gnr_workspace_bytes(ctypes.byref(p), 2, _lib.WS_FWD) == 0


def test_render_op_refuses_cpu_tensors(lib_built):
    import pytest
    import torch
    from gazenerf_amd import render, synth
    prob = synth.synth_problem(8)
    face = synth.hash_mlp_params("face")
    with pytest.raises(RuntimeError, match="no CPU path"):
        render.render_two_stream(prob["xy"], prob["R"], prob["T"], prob["Kinv"], prob["shape_code"],
                                 prob["gaze"], prob["appea_code"], face, face, n_samples=32)


def test_torch_extension_builds_and_loads(lib_built):
    """The PyTorch-ROCm C++ binding (csrc/gnr_torch.cpp) compiles against libgnr.so, imports, reports the same ABI
    and raises Python exceptions from TORCH_CHECK / gnr_last_error (no GPU needed for those paths)."""
    import pytest
    import torch
    from gazenerf_amd import _lib, _torch_ext, build
    build.build_torch_ext(verbose=False)
    ext = _torch_ext.load(required=True)
    assert ext.abi_version() == _lib.ABI_VERSION
    assert ext.saved_workspace_bytes(2, 4096, 64, 384, 258, 2) > 15e9          # cfg3: ~17 GB of saved activations
    with pytest.raises(RuntimeError, match="hidden=384"):
        ext.saved_workspace_bytes(1, 16, 64, 385, 258, 2)
    x = torch.zeros(1, 2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ext.render_fwd(x, x, x, x, x, x, x, None, None, [], [], 32, 2.5, -3.5, 384, 258, False, False, False, False, False, None, False, 0, None, None)
