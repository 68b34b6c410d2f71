"""Scratch: accuracy and timing of the N1 upsampler (HIP vs the oracle's CPU run)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import synth, neural_render
from oracle import oracle as O
dev = torch.device("cuda:0")
params = synth.hash_renderer_params(seed=2)
pd = {k: v.to(dev).requires_grad_(True) for k, v in params.items()}
for B in (1, 2):
    x = synth.synth_featmap(B, 258, 64, seed=4)
    xd = x.to(dev).requires_grad_(True)
    for mode in ("fwd", "fwdbwd"):
        ts = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.set_grad_enabled(mode == "fwdbwd"):
                img = neural_render(xd, pd)
                if mode == "fwdbwd":
                    img.square().mean().backward()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts[2:])[len(ts[2:]) // 2]
        gf = 19.6e9 * B * (3 if mode == "fwdbwd" else 1)
        print("B=%d %-7s %.3f ms  (%.1f TFLOP/s on 19.6 GFLOP/image fwd)" % (B, mode, t * 1e3, gf / t / 1e12))
torch.set_num_threads(os.cpu_count())
x = synth.synth_featmap(1, 258, 64, seed=4)
t0 = time.perf_counter()
with torch.no_grad():
    ref = O.neural_renderer(params, x, 3)
print("CPU oracle fwd B=1: %.1f ms on %d threads" % ((time.perf_counter() - t0) * 1e3, torch.get_num_threads()))
with torch.no_grad():
    got = neural_render(x.to(dev), {k: v.detach() for k, v in pd.items()})
print("image max-abs vs oracle: %.2e" % float((got.cpu() - ref).abs().max()))
