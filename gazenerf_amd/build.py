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
SOURCES = ["gnr_api.hip", "gnr_prep.hip", "gnr_fwd16.hip", "gnr_bwd.hip", "gnr_bwd16.hip", "gnr_wgrad.hip", "gnr_wgrad16.hip", "gnr_merge.hip", "gnr_vd.hip", "gnr_fwd3.hip", "gnr_bwd3.hip", "gnr_conv16.hip", "gnr_upsample.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "gnr.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# timing experiments only (tools/ab_variants.sh): extra -D switches, applied to the files named in GNR_EXTRA_FILES.  Any such
# build is compiled with -DGNR_EXPERIMENTAL_BUILD (csrc/gnr_internal.h refuses the switches without it), says so in
# gnr_build_info(), and _lib.load() refuses it unless GNR_ALLOW_EXPERIMENTAL_LIB=1 (the A/B scripts set that).
EXPERIMENT_FLAGS = os.environ.get("GNR_EXTRA_HIPCC_FLAGS", "").split()
EXPERIMENT_FILES = [f for f in os.environ.get("GNR_EXTRA_FILES", "").split(",") if f]
STAMP = os.path.join(OBJ, "build_info.txt")
# per-file extras.  gnr_wgrad.hip: the SLP vectoriser packs the scalar rider adds that sit between the MFMAs into
# v_pk_* with register shuffles around them -- each extra VALU instruction there costs matrix-pipe cycles.
EXTRA_FLAGS = {"gnr_wgrad.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_info_string() -> str:
    """What gnr_build_info() of the library being built returns: `src=<hash of csrc/ + include/gnr.h>;flags=<extra -D
    switches, with the files they were applied to>;experimental=<0|1>`."""
    from ._srchash import source_hash
    flags = ""
    if EXPERIMENT_FLAGS:
        flags = "%s@%s" % (" ".join(EXPERIMENT_FLAGS), ",".join(EXPERIMENT_FILES))
    return "src=%s;flags=%s;experimental=%d" % (source_hash(), flags, 1 if EXPERIMENT_FLAGS else 0)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    info = build_info_string()
    old_info = open(STAMP).read() if os.path.exists(STAMP) else ""
    was_experimental = "experimental=1" in old_info
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        # gnr_api.hip carries the build-info string: recompiled whenever any source (or the flag list) changed; after
        # an experimental build every object is suspect
        if (force or s in EXPERIMENT_FILES or was_experimental or _stale(obj, [src] + hdrs)
                or (s == "gnr_api.hip" and info != old_info)):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), [])
        if os.path.basename(src) in EXPERIMENT_FILES:
            cmd += EXPERIMENT_FLAGS
        if EXPERIMENT_FLAGS:
            cmd += ["-DGNR_EXPERIMENTAL_BUILD=1"]
        if os.path.basename(src) == "gnr_api.hip":
            cmd += ['-DGNR_BUILD_INFO="%s"' % info]
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
    with open(STAMP, "w") as f:
        f.write(info)
    return LIB


GUARD_SRC = os.path.join(HERE, "..", "tests", "guard", "guard_alloc.cpp")
GUARD_LIB = os.path.join(HERE, "..", "tests", "guard", "_guard_alloc.so")


def build_guard(force: bool = False, verbose: bool = True) -> str:
    """TEST INFRASTRUCTURE (like the oracle): the guard-band device allocator tests/test_guard_bands.py plugs into torch
    (tests/guard/guard_alloc.cpp: host code on the HIP runtime API; no kernels).  Built here so that it travels prebuilt."""
    if force or _stale(GUARD_LIB, [GUARD_SRC]):
        cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", GUARD_SRC, "-o", GUARD_LIB]
        if verbose:
            print("[gnr build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (GUARD_SRC, r.stdout, r.stderr))
    return os.path.abspath(GUARD_LIB)


TORCH_EXT = os.path.join(HERE, "_gnr_torch.so")


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """Compile the PyTorch-ROCm C++ binding (csrc/gnr_torch.cpp: host code only) in-tree against libgnr.so."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as E
    from ._srchash import source_hash
    src = os.path.join(CSRC, "gnr_torch.cpp")
    stamp = os.path.join(OBJ, "torch_ext_hash.txt")
    old = open(stamp).read() if os.path.exists(stamp) else ""
    if not (force or old != source_hash() or _stale(TORCH_EXT, [src, LIB, os.path.join(HERE, "..", "include", "gnr.h")])):
        return TORCH_EXT
    inc = list(E.include_paths("cuda")) + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-DTORCH_EXTENSION_NAME=_gnr_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-DUSE_ROCM=1",
           '-DGNR_SOURCE_HASH="%s"' % source_hash(),
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + i for i in inc] + [src, "-o", TORCH_EXT, "-L" + HERE, "-lgnr", "-Wl,-rpath,$ORIGIN",
                                      "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python"]
    if verbose:
        print("[gnr build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch extension build failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:]))
    with open(stamp, "w") as f:
        f.write(source_hash())
    return TORCH_EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--no-torch-ext" not in sys.argv:
        print(build_torch_ext(force="--force" in sys.argv))
    print(build_guard(force="--force" in sys.argv))
