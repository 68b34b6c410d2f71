"""Randomised shape sweep of the hot path (both precisions) against the oracle: forward outputs and a few gradients.
Not a pytest test (minutes of CPU oracle time); run by hand: python tests/diagnostics/gpu_fuzz.py [n_cases] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gazenerf_amd import render, synth
from oracle import oracle as O
torch.set_num_threads(32)
dev = torch.device("cuda:0")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
to = lambda d: {k: v.to(dev) for k, v in d.items()}
bad = 0
for case in range(n_cases):
    n_p = rng.choice([2, 7, 31, 32, 33, 64, 64, 64, 96, 128, 192, 200])
    n_rays = rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 130, 257])
    batch = rng.choice([1, 1, 2, 3])
    train = rng.random() < 0.5
    two = rng.random() < 0.7
    ds = rng.choice([1.0, 10.0, 40.0])
    seed = rng.randrange(1000)
    sub = (torch.arange(n_rays) * rng.choice([1, 7, 37, 911]) + rng.randrange(4096)) % 4096
    p = synth.synth_problem(64, batch=batch, camera=str(rng.randrange(1, 40)), seed=seed, ray_subset=sub)
    face = synth.hash_mlp_params("face", seed=seed, density_scale=ds)
    eyes = synth.hash_mlp_params("eyes", seed=seed, density_scale=ds) if two else None
    t_rand = synth.synth_jitter(batch, n_rays, n_p, seed=seed) if train else None
    edges = O.sample_edges(p["xy"], p["R"], p["T"], p["Kinv"], n_p, t_rand=t_rand)[0]
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fo = {k: v.clone().requires_grad_(True) for k, v in face.items()}
    # the oracle always evaluates two MLPs; in the single-stream cases the second one is a detached copy and only the
    # "face" outputs enter the loss
    ref = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"], leaves["appea_code"],
                              fo, eyes if two else {k: v.detach() for k, v in face.items()}, n_p, t_rand=t_rand)
    tags = ("face", "eyes") if two else ("face",)
    sum((ref["feat_" + t] ** 2).mean() + ref["bg_alpha_" + t].mean() for t in tags).backward()
    for prec in ("fp32", "bf16x3"):
        pd = to(p)
        hl = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
        hf = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
        out = render.render_two_stream(pd["xy"], hl["R"], hl["T"], pd["Kinv"], hl["shape_code"], hl["gaze"], hl["appea_code"], hf,
                                       to(eyes) if two else None, n_samples=n_p, z_edges=edges.to(dev), precision=prec)
        sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in tags).backward()
        ef = max(float((out["feat_" + t].detach().cpu() - ref["feat_" + t].detach()).abs().max()) for t in tags)
        eb = max(float((out["bg_alpha_" + t].detach().cpu() - ref["bg_alpha_" + t].detach()).abs().max()) for t in tags)
        rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
        g1 = rel(hf["RGB_layer_2.weight"].grad, fo["RGB_layer_2.weight"].grad)
        g2 = rel(hl["shape_code"].grad, leaves["shape_code"].grad)
        g3 = rel(hf["FeaExt_module_3.weight"].grad, fo["FeaExt_module_3.weight"].grad)
        ok = ef <= 1e-4 and eb <= 2.5e-4 and g1 <= 2e-2 and g2 <= 2e-2 and g3 <= 2e-2 and all(torch.isfinite(v.grad).all() for v in hl.values())
        bad += not ok
        print("%s case %2d %-6s np=%3d rays=%3d B=%d %s %s x%-4g feat %.1e bg %.1e | dRGB2 %.1e dshape %.1e dL3 %.1e" % (
            "ok  " if ok else "FAIL", case, prec, n_p, n_rays, batch, "train" if train else "test ", "2s" if two else "1s", ds, ef, eb, g1, g2, g3))
print("failures:", bad)
