"""ORACLE tooling -- test infrastructure, NOT product code.

Fixtures for SURVEY.md 8(f) N1 (NeuralRenderer + PixelShuffleUpsample + Blur).  Runs only in the build
container: imports the reference's own NeuralRenderer (models/neural_renderer.py), loads hash-generated
parameters into it, checks ``oracle.neural_renderer`` against it (forward and all gradients) and writes
tests/golden/g8_*.npz.

kornia (requirements.txt:9, ==0.6.4) is not installed here, so ``kornia.filters.filter2d`` -- used only by
Blur (pixel_shuffle_upsample.py:7-16) -- is supplied by ``oracle.kornia_filter2d``, a restatement of its
published source.  Everything else in the fixtures (1x1 convs, repeat + pixel_shuffle, bilinear upsampling,
LeakyReLU, sigmoid, their order and the autograd through them) is the reference's own code; the Blur stencil
itself is pinned only by that restatement ("Blur parity unpinned").

    python oracle/gen_golden_n1.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("GNR_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from gazenerf_amd import synth          # noqa: E402
from oracle import oracle as O          # noqa: E402


def import_reference_renderer():
    sys.dont_write_bytecode = True
    kornia = types.ModuleType("kornia")
    kfilters = types.ModuleType("kornia.filters")
    kfilters.filter2d = O.kornia_filter2d
    kornia.filters = kfilters
    sys.modules["kornia"] = kornia
    sys.modules["kornia.filters"] = kfilters
    sys.path.insert(0, REF)
    from models.neural_renderer import NeuralRenderer
    return NeuralRenderer


def run(fn, x, params):
    xg = x.clone().requires_grad_(True)
    pg = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    img = fn(xg, pg)
    # a loss with structure in every pixel and channel
    wgt = torch.linspace(0.5, 1.5, img.numel(), dtype=img.dtype).reshape(img.shape)
    ((img * wgt) ** 2).mean().backward()
    return img.detach(), xg.grad, {k: v.grad for k, v in pg.items()}


def main():
    NR = import_reference_renderer()
    cases = [("g8_upsampler_tiny", dict(feat_nc=12, featmap_size=16, img_size=64, min_feat=4), 2, 3.0),
             ("g8_upsampler_full", dict(feat_nc=258, featmap_size=64, img_size=512, min_feat=32), 1, 1.0)]
    for name, cfg, batch, wscale in cases:
        n_blocks = int(np.log2(cfg["img_size"]) - np.log2(cfg["featmap_size"]))
        print("[%s] %s, batch %d" % (name, cfg, batch))
        params = synth.hash_renderer_params(seed=2, feat_nc=cfg["feat_nc"], n_blocks=n_blocks, min_feat=cfg["min_feat"],
                                            weight_scale=wscale)
        x = synth.synth_featmap(batch, cfg["feat_nc"], cfg["featmap_size"], seed=4)
        net = NR(bg_type="white", feat_nc=cfg["feat_nc"], out_dim=3, final_actvn=True, min_feat=cfg["min_feat"],
                 featmap_size=cfg["featmap_size"], img_size=cfg["img_size"])

        def ref_fn(xx, pp):
            sd = dict(pp)
            sd["bg_featmap"] = net.bg_featmap
            return torch.func.functional_call(net, sd, (xx,))

        rimg, rdx, rdp = run(ref_fn, x, params)
        oimg, odx, odp = run(lambda xx, pp: O.neural_renderer(pp, xx, n_blocks), x, params)
        e = float((rimg - oimg).abs().max())
        print("  image max-abs %.3e" % e)
        assert e <= 1e-6, "oracle != reference (forward)"
        e = float((rdx - odx).abs().max() / rdx.abs().max())
        print("  d/dx rel %.3e" % e)
        assert e <= 1e-5
        for k in rdp:
            e = float((rdp[k] - odp[k]).abs().max() / max(float(rdp[k].abs().max()), 1e-30))
            assert e <= 1e-4, (k, e)
        arrays = {"x": x.numpy(), "out_img": rimg.numpy(), "grad_x": rdx.numpy(), "weight_seed": 2,
                  "weight_scale": wscale, "x_seed": 4, "n_blocks": n_blocks, **{"cfg_" + k: v for k, v in cfg.items()}}
        if name.endswith("full"):
            # 1.4 M parameters: keep the image on a strided pixel subset, gradients of the small tensors whole
            # and of the big ones as every 16th row
            arrays["out_img"] = rimg[:, :, ::8, ::8].numpy()
            arrays["grad_x"] = rdx[:, ::8, ::4, ::4].numpy()
            arrays.pop("x")                      # regenerated from the hash (x_seed)
            for k, g in rdp.items():
                g2 = g.reshape(g.shape[0], -1)
                arrays["gradw_" + k] = (g2[::16] if g2.numel() > 8192 else g2).numpy()
        else:
            for k, g in rdp.items():
                arrays["gradw_" + k] = g.numpy()
            for k, v in params.items():
                arrays["param_" + k] = v.numpy()
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **arrays)
        print("  wrote %s (%.0f KB)" % (os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
