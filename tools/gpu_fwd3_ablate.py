"""Time every tools/ubench/abl/libgnr_abl*.so variant of the bf16x3 kernel (one subprocess each)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
os.environ["GNR_BINDING"] = "ctypes"       # the C++ extension is linked against the in-tree libgnr.so, not the variant
sys.path.insert(0, %r)
import torch
from gazenerf_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from gazenerf_amd import render, synth
from gazenerf_amd.hiptime import KernelTimer
dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
fw, ew = to(synth.hash_mlp_params("face", seed=0, density_scale=50.0)), to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0))
p = to(synth.synth_problem(128, batch=1, seed=5))
timer = KernelTimer(); ms = []
SAVE = os.environ.get("GNR_ABL_SAVE", "0") == "1"
if SAVE:
    for d in (fw, ew):
        for v in d.values(): v.requires_grad_(True)
with torch.set_grad_enabled(SAVE):
    for i in range(6):
        with timer:
            render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], fw, ew, n_samples=64, precision=os.environ.get("GNR_ABL_PRECISION", "bf16x3"))
        if i >= 2: ms.append(timer.elapsed_ms())
t = sum(ms) / len(ms)
print("%%-28s %%.3f ms  %%.1f k rays/s  %%.0f%%%% of bf16x3 MFMA peak" %% (os.path.basename(sys.argv[1]), t, 16384 / t, 16384 * 346.03e6 / t / 1e9 / 838.9 * 100))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "tools/ubench/abl/libgnr_abl*.so")), key=lambda s: int(''.join(c for c in os.path.basename(s) if c.isdigit()) or 0)):
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr.strip()[-400:])
