"""GPU diagnostic (not collected by pytest): split invariance of the hot path on random shapes -- B images in ONE call against the
same images as two calls (per-image input gradients side by side, weight gradients added), both precisions.  Exercises workspace and
scratch sizing over (images, rays, samples): 1..200 images, 1..300 rays, 2..96 samples, hidden widths 32..384, feature widths 3..288,
in-op ray tiling, per-ray biases.  Bounds: 1e-5 / 2e-5 rel-L2 (fp32; the
summation order of a weight gradient differs between the two sides), 2e-3 for bf16x3.
    python tests/diagnostics/fuzz_hot_path_split.py [n_cases] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gazenerf_amd import render, synth        # noqa: E402


def grads(p, face, eyes, n_samples, t_rand, dev, precision, hidden=384, feat_nc=258, ray_tile=None, rb=None):
    pd = {k: v.to(dev) for k, v in p.items()}
    leaves = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
    ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
    out = render.render_two_stream(pd["xy"], leaves["R"], leaves["T"], pd["Kinv"], leaves["shape_code"], leaves["gaze"],
                                   leaves["appea_code"], fp, ep, n_samples=n_samples, t_rand=t_rand.to(dev), precision=precision,
                                   hidden=hidden, feat_nc=feat_nc, ray_tile=ray_tile,
                                   ray_bias_face=None if rb is None else rb[0].to(dev), ray_bias_eyes=None if rb is None else rb[1].to(dev))
    loss = sum((out["feat_" + t] ** 2).sum() * 1e-3 + out["bg_alpha_" + t].sum() for t in ("face", "eyes"))
    loss.backward()
    return ({k: v.grad.double().cpu() for k, v in leaves.items()}, {k: v.grad.double().cpu() for k, v in fp.items()},
            {k: v.grad.double().cpu() for k, v in ep.items()})


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    nets = {}

    def params(hidden, feat_nc):
        if (hidden, feat_nc) not in nets:
            nets[(hidden, feat_nc)] = (synth.hash_mlp_params("face", seed=6, density_scale=10.0, hidden=hidden, feat_nc=feat_nc),
                                       synth.hash_mlp_params("eyes", seed=6, density_scale=10.0, hidden=hidden, feat_nc=feat_nc))
        return nets[(hidden, feat_nc)]
    bad = 0
    for case in range(n_cases):
        B = int(rng.choice([2, 3, 5, 17, 64, 65, 66, 100, 129, 130, 200])) if rng.random() < 0.7 else int(rng.integers(2, 201))
        n_rays = int(rng.integers(1, 301)) if B <= 20 else int(rng.integers(1, 33))
        n_samples = int(rng.choice([2, 7, 16, 32, 33, 64, 96]))
        precision = "fp32" if case % 3 else "bf16x3"
        hidden = int(rng.choice([384, 384, 250, 96, 32]))
        feat_nc = int(rng.choice([258, 258, 3, 64, 200, 288]))
        face, eyes = params(hidden, feat_nc)
        ray_tile = int(rng.choice([256, 512])) if (rng.random() < 0.25 and n_rays > 256) else None      # in-op ray tiling (recompute per tile)
        rb = None
        if rng.random() < 0.3:                         # a caller-supplied per-ray bias of RGB_layer_1 (the view-direction hook)
            g = torch.Generator().manual_seed(case)
            rb = [0.1 * torch.randn(B, n_rays, hidden // 2, generator=g) for _ in range(2)]
        sub = (torch.arange(n_rays) * 251 + 5 + case) % 4096
        p = synth.synth_problem(64, batch=B, camera=str(case % 45), seed=case, ray_subset=sub)
        t_rand = synth.synth_jitter(B, n_rays, n_samples, seed=case)
        cut = int(rng.integers(1, B))
        part = lambda lo, hi: grads({k: v[lo:hi].contiguous() for k, v in p.items()}, face, eyes, n_samples, t_rand[lo:hi], dev, precision,
                                    hidden, feat_nc, ray_tile, None if rb is None else [t[lo:hi].contiguous() for t in rb])
        full, a, b = part(0, B), part(0, cut), part(cut, B)
        rel = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-30))
        e_in = max(rel(full[0][k], torch.cat([a[0][k], b[0][k]], 0)) for k in full[0])
        e_w, k_w = max((rel(full[i][k], a[i][k] + b[i][k]), k) for i in (1, 2) for k in full[i])
        tol_in, tol_w = (1e-5, 2e-5) if precision == "fp32" else (2e-3, 2e-3)
        ok = e_in <= tol_in and e_w <= tol_w and all(torch.isfinite(v).all() for d in full for v in d.values())
        bad += not ok
        print("%s %-6s B %3d (cut %3d) rays %3d samples %2d hidden %3d feat %3d tile %s bias %d: inputs %.2e  weights %.2e (%s)"
              % ("ok  " if ok else "FAIL", precision, B, cut, n_rays, n_samples, hidden, feat_nc, ray_tile, rb is not None, e_in, e_w, k_w), flush=True)
    print("%d case(s), %d violation(s)" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
