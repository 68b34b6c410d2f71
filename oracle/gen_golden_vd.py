"""ORACLE tooling -- test infrastructure, NOT product code.

Fixture for the reference's view-direction option (``include_vd=True``, models/gaze_nerf.py:70-80, 140-143, 240-243):
the sample directions pass the 4-frequency Embedder (27 channels) and enter RGB_layer_1 of both MLPs in front of the
appearance code, so that layer's weight is [192, 384 + 27 + 127].  The reference's own modules (GenSamplePoints,
Embedder x2, MLPforNeRF with vd_channels = 154, CalcRayColor) are run on 32 rays x 64 samples x B=2 in train mode and
the outputs and the gradients of the A8 loss (reference autograd) stored as ``tests/golden/g11_vd.npz``.

    python oracle/gen_golden_vd.py

Runs only where /root/reference is mounted (see oracle/gen_golden.py for the import recipe).
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from gazenerf_amd import synth                                   # noqa: E402
from oracle import oracle as O                                   # noqa: E402
from oracle import gen_golden as G                               # noqa: E402

VD_CH = 27


def main():
    torch.manual_seed(0)
    ref = G.import_reference()
    MU, MLP = ref["MU"], ref["MLPforNeRF"]
    opt = ref["BaseOptions"]({"featmap_size": 64, "featmap_nc": 258, "pred_img_size": 512})
    opt.num_sample_coarse = 64
    hidden = 384
    sub = torch.arange(0, 4096, 128) + (torch.arange(32) * 5 % 64)
    prob = synth.synth_problem(64, batch=2, camera="5", seed=13, ray_subset=sub)
    t_rand = synth.synth_jitter(2, sub.numel(), 64, seed=13)
    face = synth.hash_mlp_params("face", seed=3, vd_ch=VD_CH + synth.APPEA_DIMS, density_scale=30.0)
    eyes = synth.hash_mlp_params("eyes", seed=3, vd_ch=VD_CH + synth.APPEA_DIMS, density_scale=30.0)

    # the reference's modules, wired as GazeNeRFNet.calc_color_with_code / _forward do with include_vd=True
    sample_func = MU.GenSamplePoints(opt)
    vp_enc = MU.Embedder(N_freqs=10, include_input=True)
    vd_enc = MU.Embedder(N_freqs=4, include_input=True)                      # gaze_nerf.py:30-31, 77-79
    comp = MU.CalcRayColor()
    mlps = {tag: G.load_mlp(MLP(vp_channels=synth.VP_CH, vd_channels=synth.APPEA_DIMS + VD_CH, h_channel=hidden,
                                res_nfeat=synth.FEAT_NC), p) for tag, p in (("face", face), ("eyes", eyes))}
    leaves = {k: prob[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    with G.FixedRand(t_rand):
        sd = sample_func(prob["xy"], leaves["R"], leaves["T"], prob["Kinv"], True)
    emb, vd = vp_enc(sd["pts"]), vd_enc(sd["dirs"])                          # gaze_nerf.py:236-241
    B, _, n_r, n_p = emb.shape
    ext = torch.cat([leaves["shape_code"], leaves["gaze"]], dim=1).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, n_r, n_p)
    app = leaves["appea_code"].unsqueeze(-1).unsqueeze(-1).expand(-1, -1, n_r, n_p)
    vp_in, vd_in = torch.cat([emb, ext], dim=1), torch.cat([vd, app], dim=1)  # gaze_nerf.py:137-141
    rout = {}
    for tag in ("face", "eyes"):
        feat, sigma = mlps[tag](vp_in, vd_in)
        f, a, d, w = comp(sd["pts"], feat, sigma, sd["z_dists"], sd["zvals"])
        rout["feat_" + tag], rout["bg_alpha_" + tag] = f, a
    O.synthetic_loss(rout).backward()

    # the oracle's restatement against it
    oleaves = {k: prob[k].clone().requires_grad_(True) for k in leaves}
    ofp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in face.items())
    oep = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in eyes.items())
    oout = O.render_two_stream(prob["xy"], oleaves["R"], oleaves["T"], prob["Kinv"], oleaves["shape_code"], oleaves["gaze"],
                               oleaves["appea_code"], ofp, oep, 64, t_rand=t_rand, include_vd=True)
    O.synthetic_loss(oout).backward()
    g = dict(G.prob_arrays(prob), t_rand=t_rand, n_samples=64, weight_seed=3, density_scale=30.0, ray_subset=sub, vd_dims=VD_CH)
    for tag in ("face", "eyes"):
        for k in ("feat_", "bg_alpha_"):
            G.check("vd " + k + tag, oout[k + tag], rout[k + tag], 1e-6)
            g["out_" + k + tag] = rout[k + tag]
    for k in leaves:
        G.check("vd grad " + k, oleaves[k].grad, leaves[k].grad, 2e-6, rel=True)
        g["grad_" + k] = leaves[k].grad
    for tag, op in (("face", ofp), ("eyes", oep)):
        for name, p in mlps[tag].named_parameters():
            G.check("vd grad %s.%s" % (tag, name), op[name].grad, p.grad, 2e-6, rel=True)
            gr = p.grad
            if name == "RGB_layer_1.weight":           # all of the view-direction columns, the rest strided as in g6
                g["gradw_vdcols_%s" % tag] = gr.reshape(gr.shape[0], -1)[:, hidden:hidden + VD_CH].clone()
            if gr.numel() > 4096:
                gr = gr.reshape(gr.shape[0], -1)[::16]
            g["gradw_%s.%s" % (tag, name)] = gr
    G.save("g11_vd", **g)
    print("view-direction fixture: oracle == reference")


if __name__ == "__main__":
    main()
