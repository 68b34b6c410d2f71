"""Scratch: bf16x3 forward vs the oracle + timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import render, synth
from gazenerf_amd.hiptime import KernelTimer
from oracle import oracle as O
dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
face = synth.hash_mlp_params("face", seed=0, density_scale=50.0); eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
sub = torch.arange(0, 4096, 32) + (torch.arange(128) % 32)
p = synth.synth_problem(64, batch=1, seed=5, ray_subset=sub)
with torch.no_grad():
    ref = O.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes, 64)
    pd = to(p)
    for prec in ("fp32", "bf16x3"):
        out = render.render_two_stream(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["gaze"], pd["appea_code"],
                                       to(face), to(eyes), n_samples=64, return_depth=True, return_weights=True, precision=prec)
        torch.cuda.synchronize()
        print(prec, " ".join("%s %.2e" % (k, (out[k].cpu().double() - ref[k].double()).abs().max()) for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes", "w_face")))
timer = KernelTimer()
fw, ew = to(face), to(eyes)
for side in (128, 256):
    p = to(synth.synth_problem(side, batch=1, seed=5))
    for prec in ("fp32", "bf16x3"):
        ms = []
        with torch.no_grad():
            for i in range(5):
                with timer:
                    render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], fw, ew, n_samples=64, precision=prec)
                if i >= 2: ms.append(timer.elapsed_ms())
        t = sum(ms) / len(ms)
        print("side %d %-7s kernel %.3f ms  %.1f k rays/s  %.1f TF-equivalent" % (side, prec, t, side * side / t, side * side * 346.03e6 / t / 1e9))
