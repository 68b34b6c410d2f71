"""ORACLE tooling -- test infrastructure, NOT product code.

Pins ``oracle/oracle.py`` to the reference and writes the golden fixtures.

Runs ONLY in the build container, where the reference checkout is mounted read-only at
/root/reference.  It imports the reference's own Python modules (never copies them),
asserts that every oracle function reproduces them, and stores inputs + reference
outputs as ``tests/golden/*.npz``.  The GPU box never sees the reference; it replays the
fixtures (tests/test_oracle_golden.py, tests/test_parity_gpu.py).

    python oracle/gen_golden.py            # check + (re)write fixtures

Import-time-only third-party modules the container lacks (cv2, kornia) are replaced by
empty stand-in modules: none of their arithmetic is on the hot path.
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("GNR_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from gazenerf_amd import synth          # noqa: E402
from oracle import oracle as O          # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    kornia = types.ModuleType("kornia")
    kfilters = types.ModuleType("kornia.filters")
    kfilters.filter2d = lambda x, k, normalized=True: x       # outside the hot path
    kornia.filters = kfilters
    sys.modules.setdefault("kornia", kornia)
    sys.modules.setdefault("kornia.filters", kfilters)
    sys.path.insert(0, REF)
    os.chdir(REF)
    from configs.gazenerf_options import BaseOptions
    from models.gaze_nerf import GazeNeRFNet
    from models.mlp_nerf import MLPforNeRF
    from utils import model_utils as MU
    from utils.render_utils import RenderUtils
    return dict(BaseOptions=BaseOptions, GazeNeRFNet=GazeNeRFNet, MLPforNeRF=MLPforNeRF,
                MU=MU, RenderUtils=RenderUtils)


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())


def check(name, got, ref, tol, rel=False):
    """max-abs check; ``rel`` scales the tolerance by max|ref| (used for gradients, whose
    reduction order differs between the two autograd graphs)."""
    err = maxabs(got, ref)
    if rel:
        tol = tol * max(1.0, float(ref.abs().max()))
    status = "ok" if err <= tol else "FAIL"
    print("  %-34s max-abs %.3e (tol %.0e) %s" % (name, err, tol, status))
    if err > tol:
        raise SystemExit("oracle != reference at " + name)
    return err


def load_mlp(ref_mlp, params):
    sd = OrderedDict((k, v.clone()) for k, v in params.items())
    ref_mlp.load_state_dict(sd, strict=True)
    return ref_mlp


class FixedRand:
    """Replaces torch.rand_like for one call so the reference consumes OUR jitter."""
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.orig = torch.rand_like
        torch.rand_like = lambda x, *a, **k: self.value.to(x.dtype).reshape(x.shape)

    def __exit__(self, *exc):
        torch.rand_like = self.orig


def ref_hot_path(ref, opt, prob, face_p, eyes_p, n_samples, hidden, t_rand=None, grads=False):
    """Run the reference's own modules over the hot span gaze_nerf.py:231-162."""
    MU, MLP = ref["MU"], ref["MLPforNeRF"]
    opt.num_sample_coarse = n_samples
    sample_func = MU.GenSamplePoints(opt)
    enc = MU.Embedder(N_freqs=10, include_input=True)
    comp = MU.CalcRayColor()
    mlps = {}
    for tag, p in (("face", face_p), ("eyes", eyes_p)):
        mlps[tag] = load_mlp(MLP(vp_channels=synth.VP_CH, vd_channels=synth.APPEA_DIMS,
                                 h_channel=hidden, res_nfeat=synth.FEAT_NC), p)
    leaves = {}
    for k in ("R", "T", "shape_code", "gaze", "appea_code"):
        leaves[k] = prob[k].clone().requires_grad_(grads)
    if t_rand is None:
        sd = sample_func(prob["xy"], leaves["R"], leaves["T"], prob["Kinv"], False)
    else:
        with FixedRand(t_rand):
            sd = sample_func(prob["xy"], leaves["R"], leaves["T"], prob["Kinv"], True)
    emb = enc(sd["pts"])
    B, _, n_r, n_p = emb.shape
    shape_ext = torch.cat([leaves["shape_code"], leaves["gaze"]], dim=1)
    ext = shape_ext.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, n_r, n_p)
    app = leaves["appea_code"].unsqueeze(-1).unsqueeze(-1).expand(-1, -1, n_r, n_p)
    vp_in = torch.cat([emb, ext], dim=1)
    out = {"samples": sd, "embed": emb}
    for tag in ("face", "eyes"):
        feat, sigma = mlps[tag](vp_in, app)
        f, a, d, w = comp(sd["pts"], feat, sigma, sd["z_dists"], sd["zvals"])
        out["feat_" + tag], out["bg_alpha_" + tag], out["depth_" + tag], out["w_" + tag] = f, a, d, w
        out["feat_ps_" + tag], out["sigma_" + tag] = feat, sigma
    return out, leaves, mlps


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (np_(v) if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print("  wrote %s (%.0f KB)" % (os.path.relpath(path, ROOT), os.path.getsize(path) / 1024))


def prob_arrays(prob):
    return {"in_" + k: v for k, v in prob.items()}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    ref = import_reference()
    opt = ref["BaseOptions"]()
    MU = ref["MU"]

    # ---------------------------------------------------------------- A0: grid / intrinsics / cameras
    print("[A0] RenderUtils grid, inv_inmat, cameras")
    for side in (16, 64):
        o = ref["BaseOptions"]({"featmap_size": side, "featmap_nc": 258, "pred_img_size": 512})
        ru = ref["RenderUtils"](45, torch.device("cpu"), o)
        check("ray_xy side=%d" % side, synth.pixel_grid(side), ru.ray_xy, 0.0)
        check("inv_inmat side=%d" % side, synth.scaled_kinv(side), ru.inv_inmat, 0.0)
        r, t = synth.frontal_camera()
        check("frontal R", r, ru.base_cam_info["batch_Rmats"], 0.0)
        check("frontal T", t, ru.base_cam_info["batch_Tvecs"], 0.0)
        for v in (0, 3, 17, 44):
            r, t = synth.orbit_camera(v)
            check("orbit %d R" % v, r, ru.cam_info_list[v]["batch_Rmats"], 0.0)
            check("orbit %d T" % v, t, ru.cam_info_list[v]["batch_Tvecs"], 0.0)

    # ---------------------------------------------------------------- G1: tiny, everything committed
    print("[G1] tiny config: side 16, 8 samples, hidden 32, train-mode jitter, full grads")
    side, n_p, hidden = 16, 8, 32
    prob = synth.synth_problem(side, batch=2, camera="3", seed=11)
    face = synth.hash_mlp_params("face", seed=11, hidden=hidden, density_scale=20.0)
    eyes = synth.hash_mlp_params("eyes", seed=11, hidden=hidden, density_scale=20.0)
    t_rand = synth.synth_jitter(2, side * side, n_p, seed=11)
    rout, leaves, mlps = ref_hot_path(ref, opt, prob, face, eyes, n_p, hidden, t_rand, grads=True)
    oleaves = {k: prob[k].clone().requires_grad_(True) for k in leaves}
    ofp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in face.items())
    oep = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in eyes.items())
    oout = O.render_two_stream(prob["xy"], oleaves["R"], oleaves["T"], prob["Kinv"],
                               oleaves["shape_code"], oleaves["gaze"], oleaves["appea_code"],
                               ofp, oep, n_p, t_rand=t_rand)
    for k in ("pts", "zvals", "z_dists"):
        check("A1 " + k, oout["samples"][k], rout["samples"][k], 1e-6)
    check("A2 embed", O.embed(rout["samples"]["pts"]), rout["embed"], 1e-6)
    for tag in ("face", "eyes"):
        for k in ("feat_", "bg_alpha_", "depth_", "w_"):
            check("A4+A5 " + k + tag, oout[k + tag], rout[k + tag], 1e-6)
    O.synthetic_loss(rout).backward()
    O.synthetic_loss(oout).backward()
    g1 = dict(prob_arrays(prob), t_rand=t_rand, n_samples=n_p, hidden=hidden)
    for k in leaves:
        check("grad " + k, oleaves[k].grad, leaves[k].grad, 2e-6, rel=True)
        g1["grad_" + k] = leaves[k].grad
    for tag, op, fp in (("face", ofp, face), ("eyes", oep, eyes)):
        for name, p in mlps[tag].named_parameters():
            check("grad %s.%s" % (tag, name), op[name].grad, p.grad, 2e-6, rel=True)
            g1["gradw_%s.%s" % (tag, name)] = p.grad
            g1["w_%s.%s" % (tag, name)] = fp[name]
        for k in ("feat_", "bg_alpha_", "depth_", "w_"):
            g1["out_" + k + tag] = rout[k + tag]
    for k in ("pts", "zvals", "z_dists"):
        g1["out_" + k] = rout["samples"][k]
    save("g1_tiny", **g1)

    # A6 on the tiny config (deterministic u and a committed random u)
    print("[G1/A6] FineSample standalone")
    opt.num_sample_fine = 12
    fs_ref = MU.FineSample(opt)
    rfine = fs_ref(rout["w_face"], rout["samples"], False)
    ofine = O.fine_sample(rout["w_face"].detach(), {k: v.detach() for k, v in rout["samples"].items()}, 12)
    for k in ("pts", "zvals", "z_dists"):
        check("A6 " + k, ofine[k], rfine[k], 1e-6)
    u = torch.from_numpy(synth.hash_uniform(2 * side * side * 13, 77).astype(np.float32)).view(-1, 13)
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: u.clone()
    try:
        rfine_r = fs_ref(rout["w_face"], rout["samples"], True)
    finally:
        torch.rand = orig_rand
    ofine_r = O.fine_sample(rout["w_face"].detach(), {k: v.detach() for k, v in rout["samples"].items()}, 12, u)
    for k in ("pts", "zvals", "z_dists"):
        check("A6(rand) " + k, ofine_r[k], rfine_r[k], 1e-6)
    save("g1_fine", w_face=rout["w_face"], zvals=rout["samples"]["zvals"],
         ray_o=rout["samples"]["batch_ray_o"], ray_d=rout["samples"]["batch_ray_d"],
         ray_l=rout["samples"]["batch_ray_l"], n_fine=12, u=u,
         **{"det_" + k: rfine[k] for k in ("pts", "zvals", "z_dists")},
         **{"rnd_" + k: rfine_r[k] for k in ("pts", "zvals", "z_dists")})
    opt.num_sample_fine = 128

    # the wiring of the two streams inside the real GazeNeRFNet (hooks on CalcRayColor)
    print("[G1/net] GazeNeRFNet._forward wiring (hooked CalcRayColor outputs)")
    o16 = ref["BaseOptions"]({"featmap_size": side, "featmap_nc": 258, "pred_img_size": 128})
    o16.num_sample_coarse, o16.mlp_hidden_nchannels = n_p, hidden
    net = ref["GazeNeRFNet"](o16, False, False)
    load_mlp(net.fg_CD_predictor_face, face)
    load_mlp(net.fg_CD_predictor_eyes, eyes)
    captured = []
    h = net.calc_color_func.register_forward_hook(lambda m, i, o: captured.append(o))
    with torch.no_grad():
        net("test", prob["xy"], None, None, prob["shape_code"], prob["appea_code"], prob["gaze"],
            prob["R"], prob["T"], prob["Kinv"])
    h.remove()
    with torch.no_grad():
        otest = O.render_two_stream(prob["xy"], prob["R"], prob["T"], prob["Kinv"], prob["shape_code"],
                                    prob["gaze"], prob["appea_code"], face, eyes, n_p)
    for i, tag in enumerate(("face", "eyes")):
        check("net feat_" + tag, otest["feat_" + tag], captured[i][0], 1e-6)
        check("net bg_alpha_" + tag, otest["bg_alpha_" + tag], captured[i][1], 1e-6)

    # ---------------------------------------------------------------- G7: feature-map merge (N2)
    print("[g7_merge] GazeNeRFNet.calc_color_with_code tail: hooks on NeuralRenderer inputs")
    torch.manual_seed(3)
    net.neural_render.bg_featmap.data = torch.rand_like(net.neural_render.bg_featmap.data)
    captured, rendered = [], []
    h1 = net.calc_color_func.register_forward_hook(lambda m, i, o: captured.append(o))
    h2 = net.neural_render.register_forward_pre_hook(lambda m, i: rendered.append(i[0].detach().clone()))
    with torch.no_grad():
        net("test", prob["xy"], None, None, prob["shape_code"], prob["appea_code"], prob["gaze"],
            prob["R"], prob["T"], prob["Kinv"])
    h1.remove(); h2.remove()
    S = side
    ff, af = captured[0][0].view(2, 258, S, S), captured[0][1].view(2, 1, S, S)
    fe, ae = captured[1][0].view(2, 258, S, S), captured[1][1].view(2, 1, S, S)
    bgm = net.neural_render.bg_featmap.data
    omf, oep, om = O.merge_featmaps(ff, af, fe, ae, bgm, prob["gaze"])
    # NeuralRenderer is called on: bg_featmap, merge_face, eyes_planes, merge (gaze_nerf.py:176,200,201,205)
    check("merge_featmap_face", omf, rendered[1], 1e-6)
    check("eyes_planes", oep, rendered[2], 1e-6)
    check("merge_featmap", om, rendered[3], 1e-6)
    save("g7_merge", feat_face=ff, bg_alpha_face=af, feat_eyes=fe, bg_alpha_eyes=ae, bg_featmap=bgm,
         gaze=prob["gaze"], out_merge_face=rendered[1], out_eyes_planes=rendered[2], out_merge=rendered[3])

    # ---------------------------------------------------------------- N3: checkpoint format
    print("[ref_checkpoint_tiny] a checkpoint written the way trainer/gazenerf_trainer.py:156-191 does")
    import random as _random
    import numpy as _np
    from gazenerf_amd import checkpoint as CK
    from gazenerf_amd import HotPathRenderer
    o6 = ref["BaseOptions"]({"featmap_size": 8, "featmap_nc": 6, "pred_img_size": 32})
    o6.num_sample_coarse, o6.mlp_hidden_nchannels = 8, 32
    torch.manual_seed(5)
    net6 = ref["GazeNeRFNet"](o6, False, False)
    opt6 = torch.optim.Adam(net6.parameters(), lr=1e-4)
    state = {"torch_random_state": torch.get_rng_state(), "numpy_random_state": _np.random.get_state(),
             "random_random_state": _random.getstate(), "checkpoint_epoch": 3, "checkpoint_loss": 0.25,
             "resume_epoch": 4, "net": net6.state_dict(), "para": o6, "optimizer": opt6.state_dict(),
             "iden_offset": torch.nn.Parameter(torch.zeros(4, 100)), "expr_offset": torch.nn.Parameter(torch.zeros(4, 79)),
             "appea_offset": torch.nn.Parameter(torch.zeros(4, 127)),
             "delta_EulurAngles": torch.nn.Parameter(torch.zeros(4, 3)), "delta_Tvecs": torch.nn.Parameter(torch.zeros(4, 3, 1))}
    ck_path = os.path.join(GOLD, "ref_checkpoint_tiny.json")          # the reference names it *.json too
    torch.save(state, ck_path)
    print("  wrote %s (%.0f KB)" % (os.path.relpath(ck_path, ROOT), os.path.getsize(ck_path) / 1024))
    save("ref_checkpoint_tiny_expect", **{k: v for k, v in net6.state_dict().items() if k.startswith("fg_CD_predictor")},
         num_sample_coarse=o6.num_sample_coarse, featmap_nc=o6.featmap_nc, hidden=o6.mlp_hidden_nchannels)
    # our writer -> the reference's own resume path (torch.load + load_state_dict, gazenerf_trainer.py:102,116)
    ck = CK.load_reference_checkpoint(ck_path)
    ren = HotPathRenderer(**CK.renderer_kwargs_from_options(ck["para"]))
    missing, _ = CK.apply_to_renderer(ren, ck)
    assert not missing
    with torch.no_grad():
        ren.fg_CD_predictor_face.RGB_layer_2.bias.add_(1.0)
    CK.update_from_renderer(ck, ren)
    tmp = "/tmp/gnr_roundtrip_ckpt.json"
    CK.save_reference_checkpoint(tmp, ck)
    back = torch.load(tmp, map_location=torch.device("cpu"), weights_only=False)      # reference-side load
    assert type(back["para"]).__module__ == "configs.gazenerf_options" and back["para"].featmap_nc == 6
    net6b = ref["GazeNeRFNet"](back["para"], False, False)
    net6b.load_state_dict(back["net"])
    check("reference resumes from our checkpoint", net6b.fg_CD_predictor_face.RGB_layer_2.bias,
          net6.fg_CD_predictor_face.RGB_layer_2.bias + 1.0, 0.0)

    # ---------------------------------------------------------------- G2/G3/G4: full width
    hidden = synth.HIDDEN
    sub = torch.arange(0, 4096, 32) + (torch.arange(128) % 32)       # 128 rays, all rows/cols hit
    face = synth.hash_mlp_params("face", seed=0)
    eyes = synth.hash_mlp_params("eyes", seed=0)
    face_op = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes_op = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    cases = [("g2_np32_frontal", 32, "frontal", face, eyes, False),
             ("g2_np64_frontal", 64, "frontal", face, eyes, False),
             ("g2_np64_orbit3", 64, "3", face, eyes, False),
             ("g3_np64_train", 64, "3", face_op, eyes_op, True),
             ("g4_np64_opaque", 64, "frontal", face_op, eyes_op, False)]
    keep = None
    for name, n_p, cam, fp, ep, train in cases:
        print("[%s] side 64, %d samples, camera %s, hidden 384, 128-ray subset" % (name, n_p, cam))
        prob = synth.synth_problem(64, batch=1, camera=cam, seed=5, ray_subset=sub)
        t_rand = synth.synth_jitter(1, sub.numel(), n_p, seed=5) if train else None
        with torch.no_grad():
            rout, _, _ = ref_hot_path(ref, opt, prob, fp, ep, n_p, hidden, t_rand)
            oout = O.render_two_stream(prob["xy"], prob["R"], prob["T"], prob["Kinv"], prob["shape_code"],
                                       prob["gaze"], prob["appea_code"], fp, ep, n_p, t_rand=t_rand)
        arrays = dict(prob_arrays(prob), n_samples=n_p, weight_seed=0,
                      density_scale=(50.0 if fp is face_op else 1.0), ray_subset=sub)
        if train:
            arrays["t_rand"] = t_rand
        for tag in ("face", "eyes"):
            for k in ("feat_", "bg_alpha_", "depth_"):
                check(k + tag, oout[k + tag], rout[k + tag], 2e-6 if k != "depth_" else 2e-5)
                arrays["out_" + k + tag] = rout[k + tag]
            print("    bg_alpha_%s median %.3f min %.3f" % (tag, float(rout["bg_alpha_" + tag].median()),
                                                          float(rout["bg_alpha_" + tag].min())))
        save(name, **arrays)
        if name == "g4_np64_opaque":
            keep = (prob, rout)

    # ---------------------------------------------------------------- G5: hierarchical resampling
    print("[g5_hier] FineSample(64 -> 192) from the opaque-head face weights + fine MLP pass")
    prob, rout = keep
    fs_ref = MU.FineSample(opt)
    with torch.no_grad():
        rfine = fs_ref(rout["w_face"], rout["samples"], False)
        ofine = O.fine_sample(rout["w_face"], rout["samples"], 128)
    for k in ("pts", "zvals", "z_dists"):
        check("A6 " + k, ofine[k], rfine[k], 1e-6)
    fine_p = synth.hash_mlp_params("fine", seed=0, density_scale=50.0)
    fmlp = load_mlp(ref["MLPforNeRF"](vp_channels=synth.VP_CH, vd_channels=synth.APPEA_DIMS,
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
    check("fine feat", ofp["feat_fine"], f, 2e-6)
    check("fine bg_alpha", ofp["bg_alpha_fine"], a, 2e-6)
    save("g5_hier", **prob_arrays(prob), n_samples=64, n_fine=128, weight_seed=0, density_scale=50.0,
         w_face=rout["w_face"], out_zvals=rfine["zvals"], out_z_dists=rfine["z_dists"],
         out_pts=rfine["pts"], out_feat_fine=f, out_bg_alpha_fine=a)

    # ---------------------------------------------------------------- G6: backward, full width
    print("[g6_backward] hidden 384, 32 rays x 64 samples x B=2, train mode, grads of the A8 loss")
    sub6 = torch.arange(0, 4096, 128) + (torch.arange(32) % 64)
    prob = synth.synth_problem(64, batch=2, camera="3", seed=9, ray_subset=sub6)
    t_rand = synth.synth_jitter(2, sub6.numel(), 64, seed=9)
    rout, leaves, mlps = ref_hot_path(ref, opt, prob, face_op, eyes_op, 64, hidden, t_rand, grads=True)
    O.synthetic_loss(rout).backward()
    oleaves = {k: prob[k].clone().requires_grad_(True) for k in leaves}
    ofp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in face_op.items())
    oep = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in eyes_op.items())
    oout = O.render_two_stream(prob["xy"], oleaves["R"], oleaves["T"], prob["Kinv"], oleaves["shape_code"],
                               oleaves["gaze"], oleaves["appea_code"], ofp, oep, 64, t_rand=t_rand)
    O.synthetic_loss(oout).backward()
    g6 = dict(prob_arrays(prob), t_rand=t_rand, n_samples=64, weight_seed=0, density_scale=50.0,
              ray_subset=sub6)
    for k in leaves:
        check("grad " + k, oleaves[k].grad, leaves[k].grad, 2e-6, rel=True)
        g6["grad_" + k] = leaves[k].grad
    for tag, op in (("face", ofp), ("eyes", oep)):
        for name, p in mlps[tag].named_parameters():
            check("grad %s.%s" % (tag, name), op[name].grad, p.grad, 2e-6, rel=True)
            g = p.grad
            if g.numel() > 4096:                       # commit a strided slice of the big ones
                g = g.reshape(g.shape[0], -1)[::16]
            g6["gradw_%s.%s" % (tag, name)] = g
        for k in ("feat_", "bg_alpha_"):
            g6["out_" + k + tag] = rout[k + tag]
    save("g6_backward", **g6)
    print("all oracle-vs-reference checks passed")


if __name__ == "__main__":
    main()
