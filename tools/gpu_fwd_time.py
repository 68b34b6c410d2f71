"""Scratch: forward-only timing of the fused kernel (HIP events around the kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gazenerf_amd import render, synth
from gazenerf_amd.hiptime import KernelTimer
dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
face = to(synth.hash_mlp_params("face", seed=0, density_scale=50.0)); eyes = to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0))
timer = KernelTimer()
for side in (128, 256):
    p = to(synth.synth_problem(side, batch=1, seed=5))
    ms = []
    with torch.no_grad():
        for i in range(6):
            with timer:
                render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes, n_samples=64)
            if i >= 2: ms.append(timer.elapsed_ms())
    t = sum(ms) / len(ms)
    print("side %d: kernel %.3f ms  %.1f k rays/s  %.1f TF" % (side, t, side * side / t, side * side * 346.03e6 / t / 1e9))
