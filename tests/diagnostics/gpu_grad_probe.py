"""Scratch: per-tensor gradient errors of the HIP backward vs the oracle's autograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import render, synth
from oracle import oracle as O
dev = torch.device("cuda:0")
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n_rays, n_p, B = 40, 64, 2
sub = torch.arange(n_rays) * 53 % 4096
p = synth.synth_problem(64, batch=B, camera="9", seed=31, ray_subset=sub)
face = synth.hash_mlp_params("face", seed=4, density_scale=30.0)
eyes = synth.hash_mlp_params("eyes", seed=4, density_scale=30.0)
t_rand = synth.synth_jitter(B, n_rays, n_p, seed=6)
def run(pp, f, e, tr, fn):
    leaves = {k: pp[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.clone().requires_grad_(True) for k, v in f.items()}
    ep = {k: v.clone().requires_grad_(True) for k, v in e.items()}
    out = fn(pp["xy"], leaves["R"], leaves["T"], pp["Kinv"], leaves["shape_code"], leaves["gaze"], leaves["appea_code"], fp, ep, tr)
    loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
    loss.backward()
    return leaves, fp, ep
to = lambda d: {k: v.to(dev) for k, v in d.items()}
d64 = lambda d: {k: v.double() for k, v in d.items()}
ref = run(d64(p), d64(face), d64(eyes), t_rand.double(), lambda xy, R, T, K, s, g, a, f, e, tr: O.render_two_stream(xy, R, T, K, s, g, a, f, e, n_p, t_rand=tr))
got = run(to(p), to(face), to(eyes), t_rand.to(dev), lambda xy, R, T, K, s, g, a, f, e, tr: render.render_two_stream(xy, R, T, K, s, g, a, f, e, n_samples=n_p, t_rand=tr, precision=PREC))
torch.cuda.synchronize()
worst = [0.0]
def rep(name, g, r):
    g = g.cpu().double(); r = r.double()
    worst[0] = max(worst[0], float((g - r).norm() / max(r.norm(), 1e-30)))
    print("%-34s err %.3e  scale %.3e  rel %.2e  relL2 %.2e" % (name, (g - r).abs().max(), r.abs().max(), (g - r).abs().max() / max(r.abs().max(), 1e-30), (g - r).norm() / max(r.norm(), 1e-30)))
for k in ref[0]: rep("d" + k, got[0][k].grad, ref[0][k].grad)
for tag, i in (("face", 1), ("eyes", 2)):
    for k in ref[i]: rep(tag + "." + k, got[i][k].grad, ref[i][k].grad)
print("worst relL2 %.3e (%s)" % (worst[0], PREC))
