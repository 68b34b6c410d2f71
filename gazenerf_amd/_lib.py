"""ctypes binding of libgnr.so -- mirrors include/gnr.h field for field.

There is no CPU or PyTorch fallback: if the HIP library is missing this module raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgnr.so")

N_TRUNK = 8
N_RGB = 3
WS_FWD, WS_FWD_SAVE, WS_BWD = 0, 1, 2
STAGE_FWD_MLP, STAGE_DGRAD, STAGE_COMP_BWD, STAGE_WGRAD = 0, 1, 2, 3
N_STAGES = 4
ABI_VERSION = 4

_p = C.c_void_p


class _Sized(C.Structure):
    """Descriptor structs (since ABI 3) start with ``struct_size`` = the caller's sizeof (include/gnr.h): stamped on
    construction, checked by every entry point that takes the struct."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(type(self))


class GnrProblem(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("batch", C.c_int32), ("n_rays", C.c_int32), ("n_samples", C.c_int32),
                ("hidden", C.c_int32), ("feat_nc", C.c_int32), ("shape_dims", C.c_int32),
                ("gaze_dims", C.c_int32), ("appea_dims", C.c_int32),
                ("world_z1", C.c_float), ("world_z2", C.c_float),
                ("xy", _p), ("R", _p), ("T", _p), ("Kinv", _p),
                ("shape_code", _p), ("gaze", _p), ("appea_code", _p),
                ("t_rand", _p), ("z_edges", _p), ("edges_follow_T", C.c_int32), ("weights_packed", C.c_int32),
                ("vd_dims", C.c_int32), ("ray_bias", _p * 2)]


class GnrWeights(C.Structure):
    _fields_ = [("fea_w", _p * N_TRUNK), ("fea_b", _p * N_TRUNK),
                ("density_w", _p), ("density_b", _p),
                ("rgb_w", _p * N_RGB), ("rgb_b", _p * N_RGB)]


GnrWeightGrads = GnrWeights      # identical layout (include/gnr.h)


class GnrOutputs(C.Structure):
    _fields_ = [("feat", _p * 2), ("bg_alpha", _p * 2), ("depth", _p * 2), ("weights", _p * 2)]


class GnrOutputGrads(C.Structure):
    _fields_ = [("feat", _p * 2), ("bg_alpha", _p * 2)]


class GnrMergeProblem(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("batch", C.c_int32), ("n_pix", C.c_int32), ("feat_nc", C.c_int32),
                ("feat_face", _p), ("bg_alpha_face", _p), ("feat_eyes", _p), ("bg_alpha_eyes", _p),
                ("bg_featmap", _p), ("gaze", _p)]


UP_MAX = 4
UP_WS_FWD, UP_WS_BWD = 0, 1


class GnrUpsampleProblem(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("batch", C.c_int32), ("feat_nc", C.c_int32), ("featmap_size", C.c_int32),
                ("n_blocks", C.c_int32), ("min_feat", C.c_int32), ("final_sigmoid", C.c_int32), ("x", _p)]


class GnrUpsampleWeights(C.Structure):
    _fields_ = [("up1_w", _p * UP_MAX), ("up1_b", _p * UP_MAX), ("up2_w", _p * UP_MAX), ("up2_b", _p * UP_MAX),
                ("feat_w", _p * UP_MAX), ("feat_b", _p * UP_MAX),
                ("rgb_w", _p * (UP_MAX + 1)), ("rgb_b", _p * (UP_MAX + 1))]


GnrUpsampleWeightGrads = GnrUpsampleWeights      # identical layout (include/gnr.h)


class GnrInputGrads(C.Structure):
    _fields_ = [("R", _p), ("T", _p), ("shape_code", _p), ("gaze", _p), ("appea_code", _p), ("ray_bias", _p * 2)]


EXPORTS = ("gnr_abi_version", "gnr_build_info", "gnr_set_conv16_tile", "gnr_sizeof", "gnr_workspace_bytes", "gnr_fwd", "gnr_fwd_bf16x3", "gnr_bwd", "gnr_bwd_bf16x3", "gnr_resample",
           "gnr_sample_zvals", "gnr_set_kernel_timing", "gnr_set_aux_timing", "gnr_set_stage_timing", "gnr_set_clock_probe", "gnr_merge_scratch_bytes",
           "gnr_merge_fwd", "gnr_merge_bwd", "gnr_upsample_workspace_bytes", "gnr_upsample_fwd", "gnr_upsample_bwd",
           "gnr_last_error")

# order == the GNR_SIZEOF_* ids of include/gnr.h
STRUCTS = (GnrProblem, GnrWeights, GnrOutputs, GnrOutputGrads, GnrInputGrads, GnrMergeProblem, GnrUpsampleProblem,
           GnrUpsampleWeights)

_lib = None


class GnrError(RuntimeError):
    """Raised when a libgnr entry point returns non-zero (message from gnr_last_error)."""


def parse_build_info(info: str) -> dict:
    """gnr_build_info() -> {"src": ..., "flags": ..., "experimental": ...} (include/gnr.h)."""
    out = {"src": "unknown", "flags": "", "experimental": "0"}
    for part in info.split(";"):
        k, _, v = part.partition("=")
        if k in out:
            out[k] = v
    return out


def check_build_info(info: str, path: str):
    """A library built from other sources than the tree's, or with timing-experiment switches, must not run silently: the
    binaries are git-ignored and travel prebuilt.  `GNR_ALLOW_EXPERIMENTAL_LIB=1` admits an
    experimental build (the A/B scripts under tools/ set it); nothing admits a stale one -- rebuild."""
    import sys

    from ._srchash import source_hash
    bi = parse_build_info(info)
    want = source_hash()
    if want is not None and bi["src"] != want:
        raise RuntimeError(
            "gazenerf_amd: %s was built from other sources than this tree (library src=%s, tree src=%s): rebuild with "
            "`python -m gazenerf_amd.build`" % (path, bi["src"], want))
    if bi["experimental"] != "0" or bi["flags"]:
        if os.environ.get("GNR_ALLOW_EXPERIMENTAL_LIB", "") != "1":
            raise RuntimeError(
                "gazenerf_amd: %s is a timing-experiment build (%s): results may be wrong.  Rebuild without "
                "GNR_EXTRA_HIPCC_FLAGS, or set GNR_ALLOW_EXPERIMENTAL_LIB=1 for an A/B timing run" % (path, info))
        print("gazenerf_amd: WARNING: running an EXPERIMENTAL libgnr.so (%s)" % info, file=sys.stderr, flush=True)


def build_info() -> str:
    """gnr_build_info() of the loaded library (bench.py prints it in its result line)."""
    return load().gnr_build_info().decode("utf-8", "replace")


def load():
    """dlopen libgnr.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gazenerf_amd: %s is missing. Build it with `python -m gazenerf_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the render op." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.gnr_abi_version.restype = C.c_int
    if lib.gnr_abi_version() != ABI_VERSION:
        raise RuntimeError("libgnr.so ABI %d != binding ABI %d; rebuild with `python -m gazenerf_amd.build`" % (
            lib.gnr_abi_version(), ABI_VERSION))
    lib.gnr_build_info.restype = C.c_char_p
    check_build_info(lib.gnr_build_info().decode("utf-8", "replace"), LIB_PATH)
    lib.gnr_set_conv16_tile.restype = C.c_int
    lib.gnr_set_conv16_tile.argtypes = [C.c_int, C.c_int]
    lib.gnr_last_error.restype = C.c_char_p
    lib.gnr_workspace_bytes.restype = C.c_size_t
    lib.gnr_workspace_bytes.argtypes = [C.POINTER(GnrProblem), C.c_int, C.c_int]
    lib.gnr_fwd.restype = C.c_int
    lib.gnr_fwd.argtypes = [C.POINTER(GnrProblem), C.POINTER(GnrWeights), C.POINTER(GnrWeights),
                            C.POINTER(GnrOutputs), C.c_int, _p, C.c_size_t, _p]
    lib.gnr_fwd_bf16x3.restype = C.c_int
    lib.gnr_fwd_bf16x3.argtypes = lib.gnr_fwd.argtypes
    lib.gnr_bwd.restype = C.c_int
    lib.gnr_bwd.argtypes = [C.POINTER(GnrProblem), C.POINTER(GnrWeights), C.POINTER(GnrWeights),
                            C.POINTER(GnrOutputGrads), C.POINTER(GnrInputGrads),
                            C.POINTER(GnrWeightGrads), C.POINTER(GnrWeightGrads),
                            _p, C.c_size_t, _p, C.c_size_t, _p]
    lib.gnr_bwd_bf16x3.restype = C.c_int
    lib.gnr_bwd_bf16x3.argtypes = lib.gnr_bwd.argtypes
    lib.gnr_resample.restype = C.c_int
    lib.gnr_resample.argtypes = [_p, _p, _p, C.c_int64, C.c_int32, C.c_int32, _p, _p]
    lib.gnr_sample_zvals.restype = C.c_int
    lib.gnr_sample_zvals.argtypes = [C.POINTER(GnrProblem), _p, _p]
    lib.gnr_set_kernel_timing.restype = C.c_int
    lib.gnr_set_kernel_timing.argtypes = [_p, _p]
    lib.gnr_merge_scratch_bytes.restype = C.c_size_t
    lib.gnr_merge_scratch_bytes.argtypes = [C.POINTER(GnrMergeProblem)]
    lib.gnr_merge_fwd.restype = C.c_int
    lib.gnr_merge_fwd.argtypes = [C.POINTER(GnrMergeProblem), _p, _p, _p, _p]
    lib.gnr_upsample_workspace_bytes.restype = C.c_size_t
    lib.gnr_upsample_workspace_bytes.argtypes = [C.POINTER(GnrUpsampleProblem), C.c_int]
    lib.gnr_upsample_fwd.restype = C.c_int
    lib.gnr_upsample_fwd.argtypes = [C.POINTER(GnrUpsampleProblem), C.POINTER(GnrUpsampleWeights), _p, _p, C.c_size_t, _p]
    lib.gnr_upsample_bwd.restype = C.c_int
    lib.gnr_upsample_bwd.argtypes = [C.POINTER(GnrUpsampleProblem), C.POINTER(GnrUpsampleWeights), _p, _p,
                                     C.POINTER(GnrUpsampleWeightGrads), _p, C.c_size_t, _p, C.c_size_t, _p]
    lib.gnr_merge_bwd.restype = C.c_int
    lib.gnr_merge_bwd.argtypes = [C.POINTER(GnrMergeProblem)] + [_p] * 10 + [C.c_size_t, _p]
    lib.gnr_set_stage_timing.restype = C.c_int
    lib.gnr_set_stage_timing.argtypes = [C.c_int, _p, _p]
    lib.gnr_set_clock_probe.restype = C.c_int
    lib.gnr_set_clock_probe.argtypes = [_p]
    lib.gnr_set_aux_timing.restype = C.c_int
    lib.gnr_set_aux_timing.argtypes = [_p, _p]
    lib.gnr_sizeof.restype = C.c_size_t
    lib.gnr_sizeof.argtypes = [C.c_int]
    for which, cls in enumerate(STRUCTS):
        if lib.gnr_sizeof(which) != C.sizeof(cls):
            raise RuntimeError("libgnr.so: sizeof(%s) is %d in the library, %d in the ctypes binding" % (
                cls.__name__, lib.gnr_sizeof(which), C.sizeof(cls)))
    _lib = lib
    return lib


def check(rc: int, lib=None):
    if rc != 0:
        lib = lib or load()
        raise GnrError(lib.gnr_last_error().decode("utf-8", "replace"))
