"""Scratch probe for the GPU box: error stats vs the oracle + quick timings (not a test)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import render, synth
from oracle import oracle as O

dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
sub = torch.arange(0, 4096, 32) + (torch.arange(128) % 32)
p = synth.synth_problem(64, batch=1, seed=5, ray_subset=sub)
with torch.no_grad():
    ref = O.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes, 64)
    pd = to(p)
    out = render.render_two_stream(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["gaze"], pd["appea_code"],
                                   to(face), to(eyes), n_samples=64, return_depth=True, return_weights=True)
    torch.cuda.synchronize()
for k in ("feat_face", "bg_alpha_face", "depth_face", "w_face", "feat_eyes", "bg_alpha_eyes"):
    d = (out[k].cpu().double() - ref[k].double()).abs()
    print("%-14s max-abs err %.3e   (ref max %.3e)" % (k, d.max(), ref[k].abs().max()))

for side, reps in ((64, 10), (128, 5), (512, 2)):
    p = to(synth.synth_problem(side, batch=1, seed=5))
    fw, ew = to(face), to(eyes)
    with torch.no_grad():
        for _ in range(2):
            render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], fw, ew, n_samples=64)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(reps):
            render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], fw, ew, n_samples=64)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
    rays = side * side
    print("side %4d: %.3f ms/img  %.1f k rays/s   %.1f TFLOP/s (folded count)" % (side, dt * 1e3, rays / dt / 1e3, rays / dt * 346.03e6 / 1e12))
