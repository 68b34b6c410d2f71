"""ORACLE tooling -- test infrastructure, NOT product code.

Side-512 fixtures: the geometry of BASELINE cfg2b / cfg5 (``featmap_size=512``,
configs/gazenerf_options.py:29-35; inv_inmat focal terms divided by 512/32, utils/render_utils.py:36-40;
pixel coordinates up to 511).  The reference's own modules are run on a strided 256-ray subset of the
512x512 grid and the outputs stored as ``tests/golden/g2b_*.npz`` / ``g5b_hier512.npz`` / ``g6b_backward512.npz``.

    python oracle/gen_golden_s512.py

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

SIDE = 512


def ray_subset_512(n=256):
    """n rays of the 512x512 grid hitting every 2nd row band and all column residues, incl. the four
    corners (x or y = 0 / 511: the extreme ray directions)."""
    idx = (torch.arange(n - 4, dtype=torch.int64) * 1039 + 17) % (SIDE * SIDE)     # 1039 prime: rows and columns mix
    corners = torch.tensor([0, SIDE - 1, SIDE * (SIDE - 1), SIDE * SIDE - 1])
    return torch.cat([corners, idx])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    ref = G.import_reference()
    opt = ref["BaseOptions"]()
    MU = ref["MU"]
    hidden = synth.HIDDEN

    print("[A0] RenderUtils at featmap_size=512")
    o = ref["BaseOptions"]({"featmap_size": SIDE, "featmap_nc": 258, "pred_img_size": 512})
    ru = ref["RenderUtils"](45, torch.device("cpu"), o)
    G.check("ray_xy side=512", synth.pixel_grid(SIDE), ru.ray_xy, 0.0)
    G.check("inv_inmat side=512", synth.scaled_kinv(SIDE), ru.inv_inmat, 0.0)

    sub = ray_subset_512()
    face = synth.hash_mlp_params("face", seed=0)
    eyes = synth.hash_mlp_params("eyes", seed=0)
    face_op = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes_op = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    cases = [("g2b_np64_frontal", "frontal", face, eyes, False),
             ("g2b_np64_orbit3", "3", face, eyes, False),
             ("g2b_np64_opaque", "frontal", face_op, eyes_op, False),
             ("g2b_np64_train_opaque", "3", face_op, eyes_op, True)]
    keep = None
    for name, cam, fp, ep, train in cases:
        print("[%s] side 512, 64 samples, camera %s, hidden 384, %d-ray subset" % (name, cam, sub.numel()))
        prob = synth.synth_problem(SIDE, batch=1, camera=cam, seed=5, ray_subset=sub)
        t_rand = synth.synth_jitter(1, sub.numel(), 64, seed=15) if train else None
        with torch.no_grad():
            rout, _, _ = G.ref_hot_path(ref, opt, prob, fp, ep, 64, hidden, t_rand)
            oout = O.render_two_stream(prob["xy"], prob["R"], prob["T"], prob["Kinv"], prob["shape_code"],
                                       prob["gaze"], prob["appea_code"], fp, ep, 64, t_rand=t_rand)
        arrays = dict(G.prob_arrays(prob), n_samples=64, weight_seed=0, side=SIDE,
                      density_scale=(50.0 if fp is face_op else 1.0), ray_subset=sub)
        if train:
            arrays["t_rand"] = t_rand
        for tag in ("face", "eyes"):
            for k in ("feat_", "bg_alpha_", "depth_"):
                G.check(k + tag, oout[k + tag], rout[k + tag], 2e-6 if k != "depth_" else 2e-5)
                arrays["out_" + k + tag] = rout[k + tag]
            print("    bg_alpha_%s median %.3f min %.3f" % (tag, float(rout["bg_alpha_" + tag].median()),
                                                          float(rout["bg_alpha_" + tag].min())))
        arrays["out_zvals"] = rout["samples"]["zvals"]
        G.save(name, **arrays)
        if name == "g2b_np64_opaque":
            keep = (prob, rout)

    print("[g5b_hier512] FineSample(64 -> 192) at side 512 from the opaque-head face weights + fine MLP pass")
    prob, rout = keep
    opt.num_sample_fine = 128
    fs_ref = MU.FineSample(opt)
    with torch.no_grad():
        rfine = fs_ref(rout["w_face"], rout["samples"], False)
        ofine = O.fine_sample(rout["w_face"], rout["samples"], 128)
    for k in ("pts", "zvals", "z_dists"):
        G.check("A6 " + k, ofine[k], rfine[k], 1e-6)
    fine_p = synth.hash_mlp_params("fine", seed=0, density_scale=50.0)
    fmlp = G.load_mlp(ref["MLPforNeRF"](vp_channels=synth.VP_CH, vd_channels=synth.APPEA_DIMS,
                                        h_channel=hidden, res_nfeat=synth.FEAT_NC), fine_p)
    with torch.no_grad():
        emb = MU.Embedder(N_freqs=10, include_input=True)(rfine["pts"])
        B, _, n_r, n_p = emb.shape
        ext = torch.cat([prob["shape_code"], prob["gaze"]], 1).view(B, -1, 1, 1).expand(-1, -1, n_r, n_p)
        app = prob["appea_code"].view(B, -1, 1, 1).expand(-1, -1, n_r, n_p)
        feat, sigma = fmlp(torch.cat([emb, ext], 1), app)
        f, a, d, w = MU.CalcRayColor()(rfine["pts"], feat, sigma, rfine["z_dists"], rfine["zvals"])
        ofp = O.hier_fine_pass({"w_face": rout["w_face"], "samples": rout["samples"]}, prob["shape_code"],
                               prob["gaze"], prob["appea_code"], fine_p, 128)
    G.check("fine feat", ofp["feat_fine"], f, 2e-6)
    G.check("fine bg_alpha", ofp["bg_alpha_fine"], a, 2e-6)
    G.save("g5b_hier512", **G.prob_arrays(prob), n_samples=64, n_fine=128, weight_seed=0, density_scale=50.0,
           side=SIDE, ray_subset=sub, w_face=rout["w_face"], out_zvals=rfine["zvals"],
           out_z_dists=rfine["z_dists"], out_feat_fine=f, out_bg_alpha_fine=a)

    print("[g6b_backward512] side 512, 32 rays x 64 samples x B=2, train mode, grads of the A8 loss")
    sub6 = torch.cat([torch.tensor([0, SIDE * SIDE - 1]), (torch.arange(30) * 8737 + 511) % (SIDE * SIDE)])
    prob = synth.synth_problem(SIDE, batch=2, camera="3", seed=19, ray_subset=sub6)
    t_rand = synth.synth_jitter(2, sub6.numel(), 64, seed=19)
    rout, leaves, mlps = G.ref_hot_path(ref, opt, prob, face_op, eyes_op, 64, hidden, t_rand, grads=True)
    O.synthetic_loss(rout).backward()
    oleaves = {k: prob[k].clone().requires_grad_(True) for k in leaves}
    ofp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in face_op.items())
    oep = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in eyes_op.items())
    oout = O.render_two_stream(prob["xy"], oleaves["R"], oleaves["T"], prob["Kinv"], oleaves["shape_code"],
                               oleaves["gaze"], oleaves["appea_code"], ofp, oep, 64, t_rand=t_rand)
    O.synthetic_loss(oout).backward()
    g6 = dict(G.prob_arrays(prob), t_rand=t_rand, n_samples=64, weight_seed=0, density_scale=50.0,
              side=SIDE, ray_subset=sub6)
    for k in leaves:
        G.check("grad " + k, oleaves[k].grad, leaves[k].grad, 2e-6, rel=True)
        g6["grad_" + k] = leaves[k].grad
    for tag, op in (("face", ofp), ("eyes", oep)):
        for name, p in mlps[tag].named_parameters():
            G.check("grad %s.%s" % (tag, name), op[name].grad, p.grad, 2e-6, rel=True)
            g = p.grad
            if g.numel() > 4096:
                g = g.reshape(g.shape[0], -1)[::16]
            g6["gradw_%s.%s" % (tag, name)] = g
        for k in ("feat_", "bg_alpha_"):
            g6["out_" + k + tag] = rout[k + tag]
    G.save("g6b_backward512", **g6)
    print("all side-512 oracle-vs-reference checks passed")


if __name__ == "__main__":
    main()
