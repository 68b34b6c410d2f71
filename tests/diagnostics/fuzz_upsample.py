"""GPU diagnostic (not collected by pytest): the upsampler against the CPU oracle on random shapes -- channel counts 3..300 (with the
GEMM routing's special cases 32 / 64 / 128 / 129 / 256 / 258 over-represented), 1-3 blocks from 16 x 16 / 32 x 32 maps, 1-24 stacked
maps.  Prints image max-abs and the worst gradient rel-L2 per case; exits non-zero on a violation of the test suite's bounds
(1e-4 / 1e-3).      python tests/diagnostics/fuzz_upsample.py [n_cases] [seed]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gazenerf_amd import neural_render, synth        # noqa: E402
from oracle import oracle as O                      # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    special = [32, 64, 128, 129, 256, 258, 48, 96, 100]
    bad = 0
    for case in range(n_cases):
        n_blocks = int(rng.integers(1, 4))
        side = int(rng.choice([16, 32]))
        if side << n_blocks > 128:
            n_blocks = 2 if side == 32 else 3
        feat_nc = int(rng.choice(special)) if rng.random() < 0.6 else int(rng.integers(3, 301))
        min_feat = int(rng.integers(1, feat_nc + 1)) if rng.random() < 0.5 else max(1, feat_nc >> int(rng.integers(0, 4)))
        batch = int(rng.integers(1, 25)) if feat_nc <= 129 else int(rng.integers(1, 9))
        params = synth.hash_renderer_params(seed=100 + case, feat_nc=feat_nc, n_blocks=n_blocks, min_feat=min_feat, weight_scale=2.0)
        x = synth.synth_featmap(batch, feat_nc, side, seed=case)
        xg = x.clone().requires_grad_(True)
        pg = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        ref = O.neural_renderer(pg, xg, n_blocks)
        w = torch.linspace(0.5, 1.5, ref[0].numel()).reshape(ref[0].shape)
        (ref * w).sum().backward()
        xd = x.to(dev).clone().requires_grad_(True)
        pd = {k: v.to(dev).clone().requires_grad_(True) for k, v in params.items()}
        img = neural_render(xd, pd, n_blocks=n_blocks, min_feat=min_feat)
        (img * w.to(dev)).sum().backward()
        e_img = float((img.detach().cpu() - ref.detach()).abs().max())
        worst = max([("x", rel_l2(xd.grad.cpu(), xg.grad))] + [(k, rel_l2(pd[k].grad.cpu(), pg[k].grad)) for k in pg], key=lambda t: t[1])
        ok = e_img <= 1e-4 and worst[1] <= 1e-3
        bad += not ok
        print("%s feat_nc %3d min_feat %3d side %2d blocks %d batch %2d: image %.2e  worst gradient %.2e (%s)"
              % ("ok  " if ok else "FAIL", feat_nc, min_feat, side, n_blocks, batch, e_img, worst[1], worst[0]), flush=True)
    print("%d case(s), %d violation(s)" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
