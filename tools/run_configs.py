"""BASELINE.json configs on one MI355X (beyond the bench's cfg2b): wall time per call, both precisions.

  cfg2a  64x64 rays, 64 samples, B=1, forward              (what the reference does per 512x512 image)
  cfg3   64x64 rays, 64 samples, B=2, train jitter, forward + backward of the A8 loss
  cfg3n  the same step through the whole network (hot path -> merge -> upsampler x4 -> image loss)
  cfg5   512x512 rays: coarse 64 (face + eyes) -> FineSample -> 192-sample fine pass through a third MLP

usage: python tools/run_configs.py [--reps 5]      (run under rocprofv3 for the kernel split)
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gazenerf_amd import render, synth, GazeNeRFNetAMD

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
to = lambda d: {k: v.to(dev) for k, v in d.items()}
face, eyes = to(synth.hash_mlp_params("face", seed=0, density_scale=50.0)), to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0))
fine = to(synth.hash_mlp_params("fine", seed=0, density_scale=50.0))


def timed(fn):
    ts = []
    for i in range(args.reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2] * 1e3


rows = []
for prec in ("fp32", "bf16x3"):
    # cfg2a
    p = to(synth.synth_problem(64, batch=1, seed=5))
    def cfg2a():
        with torch.no_grad():
            render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes,
                                     n_samples=64, precision=prec)
    t = timed(cfg2a); rows.append(("cfg2a fwd 4096 rays", prec, t, 4096 / t))
    wc = render.PackedWeightCache()
    def cfg2a_cached():
        with torch.no_grad():
            render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes,
                                     n_samples=64, precision=prec, weight_cache=wc)
    t = timed(cfg2a_cached); rows.append(("cfg2a fwd, packed weights cached", prec, t, 4096 / t))
    # cfg3
    p3 = to(synth.synth_problem(64, batch=2, camera="3", seed=7))
    tr = synth.synth_jitter(2, 4096, 64, seed=1).to(dev)
    params = [v.clone().requires_grad_(True) for v in list(face.values()) + list(eyes.values())]
    fp = dict(zip(face.keys(), params[:24])); ep = dict(zip(eyes.keys(), params[24:]))
    leaves = {k: p3[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    def cfg3():
        out = render.render_two_stream(p3["xy"], leaves["R"], leaves["T"], p3["Kinv"], leaves["shape_code"], leaves["gaze"],
                                       leaves["appea_code"], fp, ep, n_samples=64, t_rand=tr, precision=prec)
        loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
        loss.backward()
    t = timed(cfg3); rows.append(("cfg3 fwd+bwd 2x4096 rays", prec, t, 8192 / t))
    # cfg3 through the whole network
    net = GazeNeRFNetAMD(precision=prec).to(dev)
    target = torch.rand(2, 3, 512, 512, device=dev)
    def cfg3n():
        net.zero_grad(set_to_none=True)
        res = net("train", p3["xy"], None, None, p3["shape_code"], p3["appea_code"], p3["gaze"], p3["R"], p3["T"], p3["Kinv"],
                  t_rand=tr)["coarse_dict"]
        sum(((res[k] - target) ** 2).mean() for k in res).backward()
    t = timed(cfg3n); rows.append(("cfg3n whole-network step 2 images", prec, t, 8192 / t))
    # cfg5
    p5 = to(synth.synth_problem(512, batch=1, seed=9))
    def cfg5():
        with torch.no_grad():
            c = render.render_two_stream(p5["xy"], p5["R"], p5["T"], p5["Kinv"], p5["shape_code"], p5["gaze"], p5["appea_code"], face, eyes,
                                         n_samples=64, return_weights=True, precision=prec)
            zv = render.sample_zvals(p5["xy"], p5["R"], p5["T"], p5["Kinv"], n_samples=64)
            edges = render.importance_resample(c["w_face"], zv, n_fine=128)
            render.render_two_stream(p5["xy"], p5["R"], p5["T"], p5["Kinv"], p5["shape_code"], p5["gaze"], p5["appea_code"], fine, None,
                                     n_samples=192, z_edges=edges, precision=prec)
    t = timed(cfg5); rows.append(("cfg5 hier 262144 rays (64x2 + 192x1)", prec, t, 262144 / t))
print("%-40s %-7s %12s %14s" % ("config", "prec", "ms / call", "k rays/s"))
for name, prec, t, r in rows:
    print("%-40s %-7s %12.2f %14.1f" % (name, prec, t, r))
