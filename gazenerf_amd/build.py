"""Build libgnr.so (the C-ABI HIP library) in-tree for gfx950.

    python -m gazenerf_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to gazenerf_amd/csrc/build/, the library to
gazenerf_amd/libgnr.so (git-ignored, but it travels with the tree to the GPU box).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libgnr.so")
SOURCES = ["gnr_api.hip", "gnr_prep.hip", "gnr_fwd.hip", "gnr_fwd16.hip", "gnr_bwd.hip", "gnr_bwd16.hip", "gnr_wgrad.hip", "gnr_merge.hip", "gnr_vd.hip", "gnr_fwd3.hip", "gnr_bwd3.hip", "gnr_conv16.hip", "gnr_upsample.hip"]
HEADERS = ["gnr_internal.h", "gnr_device.h", "gnr_chain.h", "gnr_chain16.h", "gnr_conv16.h", "gnr_chain3.h", "gnr_bwd_common.h", os.path.join("..", "..", "include", "gnr.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# timing experiments only (tools/ab_variants.sh): extra -D switches, applied to the files named in GNR_EXTRA_FILES
EXPERIMENT_FLAGS = os.environ.get("GNR_EXTRA_HIPCC_FLAGS", "").split()
EXPERIMENT_FILES = [f for f in os.environ.get("GNR_EXTRA_FILES", "").split(",") if f]
# per-file extras.  gnr_wgrad.hip: the SLP vectoriser packs the scalar rider adds that sit between the MFMAs into
# v_pk_* with register shuffles around them -- each extra VALU instruction there costs matrix-pipe cycles.
EXTRA_FLAGS = {"gnr_wgrad.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or s in EXPERIMENT_FILES or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])
        if os.path.basename(src) in EXPERIMENT_FILES:
            cmd += EXPERIMENT_FLAGS
        cmd += ["-c", src, "-o", obj]
        if verbose:
            print("[gnr build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return r.stderr

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for warn in ex.map(compile_one, jobs):
            if warn.strip() and verbose:
                print(warn)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[gnr build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


TORCH_EXT = os.path.join(HERE, "_gnr_torch.so")


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """Compile the PyTorch-ROCm C++ binding (csrc/gnr_torch.cpp: host code only) in-tree against libgnr.so."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as E
    src = os.path.join(CSRC, "gnr_torch.cpp")
    if not (force or _stale(TORCH_EXT, [src, LIB, os.path.join(HERE, "..", "include", "gnr.h")])):
        return TORCH_EXT
    inc = list(E.include_paths("cuda")) + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-DTORCH_EXTENSION_NAME=_gnr_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + i for i in inc] + [src, "-o", TORCH_EXT, "-L" + HERE, "-lgnr", "-Wl,-rpath,$ORIGIN",
                                      "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python"]
    if verbose:
        print("[gnr build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch extension build failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:]))
    return TORCH_EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--no-torch-ext" not in sys.argv:
        print(build_torch_ext(force="--force" in sys.argv))
