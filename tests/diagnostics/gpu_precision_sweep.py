"""Scratch: max-abs error of feat / bg_alpha vs the fp64 oracle over a sweep of stress problems, both precisions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import render, synth
from oracle import oracle as O
torch.set_num_threads(32)
dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
d64 = lambda d: {k: v.double() for k, v in d.items()}
worst = {}
for ds in (1.0, 10.0, 50.0):
    for seed in range(4):
        n_rays = 48
        p = synth.synth_problem(64, batch=2, camera=str(1 + 3 * seed), seed=seed, ray_subset=torch.arange(n_rays) * (85 + seed) % 4096)
        face = synth.hash_mlp_params("face", seed=seed, density_scale=ds); eyes = synth.hash_mlp_params("eyes", seed=seed, density_scale=ds)
        t_rand = synth.synth_jitter(2, n_rays, 64, seed=seed)
        with torch.no_grad():
            ref = O.render_two_stream(*[p[k].double() for k in ("xy", "R", "T", "Kinv", "shape_code", "gaze", "appea_code")], d64(face), d64(eyes), 64, t_rand=t_rand.double())
            r32 = O.render_two_stream(*[p[k] for k in ("xy", "R", "T", "Kinv", "shape_code", "gaze", "appea_code")], face, eyes, 64, t_rand=t_rand)
            pd = to(p)
            outs = {"ref32": r32}
            for prec in ("fp32", "bf16x3"):
                outs[prec] = render.render_two_stream(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["gaze"], pd["appea_code"],
                                                      to(face), to(eyes), n_samples=64, t_rand=t_rand.to(dev), precision=prec)
        for name, o in outs.items():
            for k in ("feat_face", "feat_eyes", "bg_alpha_face", "bg_alpha_eyes"):
                e = float((o[k].cpu().double() - ref[k]).abs().max())
                key = (ds, name, k.split("_")[0] if k.startswith("feat") else "bg_alpha")
                worst[key] = max(worst.get(key, 0.0), e)
print("max-abs vs the fp64 oracle (4 problems each): density_scale, path, output")
for k in sorted(worst):
    print("  x%-4g %-7s %-9s %.2e" % (k[0], k[1], k[2], worst[k]))
