"""GPU: the HIP path (through the C ABI) against the committed reference fixtures and against the
oracle run live on the same seeded inputs.

Tolerance (BASELINE.json north_star): 1e-4 max-abs on the low-res feature map and on bg_alpha.
Depth carries its own fp32 noise floor (sum w*z with z up to 15.4; SURVEY.md 8(c) measured
8e-5..6e-4 between fp32 and fp64 runs of the reference itself), so depth uses 1e-4 * 15.4.
"""
import pytest
import torch

from conftest import golden_problem, load_golden
from gazenerf_amd import render, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEPTH_TOL = 1e-4 * 15.4
# per-sample weights w_i are not part of the north-star bound; with the x40 "opaque head" and few,
# thick samples (N_p = 8: delta = 0.75) an fp32 ulp in pts (x512 in the top encoding frequency) moves
# alpha_i by ~1e-4 while the composited feature map stays within 1e-4.
W_TOL = 3e-4


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _to(d, dev):
    return {k: v.to(dev) for k, v in d.items()}


def _maxabs(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def _weights(g):
    ds, seed = float(g["density_scale"]), int(g["weight_seed"])
    return (synth.hash_mlp_params("face", seed=seed, density_scale=ds),
            synth.hash_mlp_params("eyes", seed=seed, density_scale=ds))


def _hip(p, face, eyes, n_samples, dev, **kw):
    pd = _to(p, dev)
    for k in ("t_rand", "z_edges"):
        if kw.get(k) is not None:
            kw[k] = kw[k].to(dev)
    return render.render_two_stream(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["gaze"],
                                    pd["appea_code"], _to(face, dev), _to(eyes, dev) if eyes else None,
                                    n_samples=n_samples, **kw)


PRECISIONS = ["fp32", "bf16x3"]   # bf16x3: gnr_fwd_bf16x3, the split-bf16 inference kernel -- same bound


# g2b_*: the same checks on cfg2b geometry -- featmap_size=512, pixel coordinates up to 511, focal terms / 16
# (configs/gazenerf_options.py:29-35, utils/render_utils.py:36-40); fixtures from oracle/gen_golden_s512.py
S512_FIXTURES = ["g2b_np64_frontal", "g2b_np64_orbit3", "g2b_np64_opaque", "g2b_np64_train_opaque"]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["g2_np32_frontal", "g2_np64_frontal", "g2_np64_orbit3",
                                  "g3_np64_train", "g4_np64_opaque"] + S512_FIXTURES)
def test_forward_vs_reference_fixture(name, precision):
    dev = _dev()
    g = load_golden(name)
    face, eyes = _weights(g)
    with torch.no_grad():
        out = _hip(golden_problem(g), face, eyes, int(g["n_samples"]), dev, t_rand=g.get("t_rand"),
                   return_depth=True, precision=precision)
    for tag in ("face", "eyes"):
        assert _maxabs(out["feat_" + tag], g["out_feat_" + tag]) <= TOL
        assert _maxabs(out["bg_alpha_" + tag], g["out_bg_alpha_" + tag]) <= TOL
        assert _maxabs(out["depth_" + tag], g["out_depth_" + tag]) <= DEPTH_TOL
    if "out_zvals" in g:
        pd = _to(golden_problem(g), dev)
        tr = g.get("t_rand")
        zv = render.sample_zvals(pd["xy"], pd["R"], pd["T"], pd["Kinv"], n_samples=int(g["n_samples"]),
                                 t_rand=tr.to(dev) if tr is not None else None)
        assert _maxabs(zv, g["out_zvals"]) <= 4e-6


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n_samples,n_rays,batch,train", [(64, 96, 2, True), (32, 50, 1, False),
                                                          (40, 33, 1, True), (192, 16, 1, False),
                                                          (8, 7, 3, True)])
def test_forward_vs_oracle_live(n_samples, n_rays, batch, train, precision):
    """Ragged sizes: ray counts that do not fill a workgroup, sample counts that do not fill a
    32-sample chunk (40, 8) and the 192-sample fine-pass width.

    For step counts that are not a power of two, torch.linspace's vectorised CPU kernel rounds
    differently from a scalar evaluation (and differently per host ISA), so the plane-sweep edges
    themselves differ by an ulp; an opaque density head amplifies that in the per-sample weights.
    Those sizes are therefore checked twice: with the oracle's own edges passed explicitly
    (exact same z -> full 1e-4 on everything) and through the built-in sweep with 5e-4 on the
    per-sample weights only.  The reference's own sample counts (32, 64) are powers of two."""
    dev = _dev()
    sub = torch.arange(n_rays) * 37 % 4096
    p = synth.synth_problem(64, batch=batch, camera="5", seed=21, ray_subset=sub)
    face = synth.hash_mlp_params("face", seed=2, density_scale=40.0)
    eyes = synth.hash_mlp_params("eyes", seed=2, density_scale=40.0)
    t_rand = synth.synth_jitter(batch, n_rays, n_samples, seed=4) if train else None
    pow2 = (n_samples & (n_samples - 1)) == 0
    with torch.no_grad():
        ref = O.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                  p["appea_code"], face, eyes, n_samples, t_rand=t_rand)
        out = _hip(p, face, eyes, n_samples, dev, t_rand=t_rand, return_depth=True, return_weights=True,
                   precision=precision)
        edges = O.sample_edges(p["xy"], p["R"], p["T"], p["Kinv"], n_samples, t_rand=t_rand)[0]
        out_e = _hip(p, face, eyes, n_samples, dev, z_edges=edges, return_depth=True, return_weights=True,
                     precision=precision)
    for tag in ("face", "eyes"):
        for o, wtol in ((out, W_TOL if pow2 else 5 * TOL), (out_e, W_TOL)):
            assert _maxabs(o["feat_" + tag], ref["feat_" + tag]) <= TOL
            assert _maxabs(o["bg_alpha_" + tag], ref["bg_alpha_" + tag]) <= TOL
            assert _maxabs(o["w_" + tag], ref["w_" + tag]) <= wtol
            assert _maxabs(o["depth_" + tag], ref["depth_" + tag]) <= DEPTH_TOL
    zv = render.sample_zvals(p["xy"].to(dev), p["R"].to(dev), p["T"].to(dev), p["Kinv"].to(dev),
                             n_samples=n_samples, t_rand=t_rand.to(dev) if train else None)
    assert _maxabs(zv, ref["samples"]["zvals"]) <= 4e-6


def test_bf16x3_is_as_close_to_exact_as_the_reference_fp32():
    """Stress case (x50 opaque density head, train jitter, orbit cameras): against the oracle in fp64, the bf16x3
    kernels are no further from the exact result than the reference's own fp32 arithmetic (oracle in fp32) -- on
    bg_alpha both are ~1.5e-4 away, so they may differ from EACH OTHER by about that much; the feature map stays
    within 1e-4 of the reference's fp32 output everywhere."""
    dev = _dev()
    worst = {}
    for seed in range(3):
        n_rays = 40
        p = synth.synth_problem(64, batch=2, camera=str(1 + 3 * seed), seed=seed,
                                ray_subset=torch.arange(n_rays) * (85 + seed) % 4096)
        face = synth.hash_mlp_params("face", seed=seed, density_scale=50.0)
        eyes = synth.hash_mlp_params("eyes", seed=seed, density_scale=50.0)
        t_rand = synth.synth_jitter(2, n_rays, 64, seed=seed)
        d64 = lambda d: {k: v.double() for k, v in d.items()}
        keys = ("xy", "R", "T", "Kinv", "shape_code", "gaze", "appea_code")
        with torch.no_grad():
            exact = O.render_two_stream(*[p[k].double() for k in keys], d64(face), d64(eyes), 64, t_rand=t_rand.double())
            ref32 = O.render_two_stream(*[p[k] for k in keys], face, eyes, 64, t_rand=t_rand)
            x3 = _hip(p, face, eyes, 64, dev, t_rand=t_rand, precision="bf16x3")
        for k in ("feat_face", "feat_eyes", "bg_alpha_face", "bg_alpha_eyes"):
            kind = "feat" if k.startswith("feat") else "bg_alpha"
            e_ref = float((ref32[k].double() - exact[k]).abs().max())
            e_x3 = float((x3[k].cpu().double() - exact[k]).abs().max())
            w = worst.setdefault(kind, [0.0, 0.0, 0.0])
            w[0], w[1] = max(w[0], e_ref), max(w[1], e_x3)
            w[2] = max(w[2], _maxabs(x3[k], ref32[k]))
    for kind, (e_ref, e_x3, diff) in worst.items():
        assert e_x3 <= 1.25 * e_ref + 1e-5, (kind, e_ref, e_x3)
    assert worst["feat"][2] <= TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_single_stream_equals_two_stream(precision):
    dev = _dev()
    p = synth.synth_problem(64, batch=1, seed=3, ray_subset=torch.arange(64))
    face = synth.hash_mlp_params("face", seed=1, density_scale=30.0)
    eyes = synth.hash_mlp_params("eyes", seed=1, density_scale=30.0)
    with torch.no_grad():
        two = _hip(p, face, eyes, 64, dev, precision=precision)
        one = _hip(p, eyes, None, 64, dev, precision=precision)
    assert torch.equal(two["feat_eyes"], one["feat_face"])
    assert torch.equal(two["bg_alpha_eyes"], one["bg_alpha_face"])


@pytest.mark.parametrize("name", ["g5_hier", "g5b_hier512"])
def test_hierarchical_vs_reference_fixture(name):
    """cfg5: coarse 64 -> FineSample -> 192-sample fine pass through a third MLP; g5b is the
    same at featmap_size=512, the geometry BASELINE cfg5 is quoted on."""
    dev = _dev()
    g = load_golden(name)
    p = golden_problem(g)
    face, eyes = _weights(g)
    fine = synth.hash_mlp_params("fine", seed=int(g["weight_seed"]), density_scale=float(g["density_scale"]))
    pd = _to(p, dev)
    with torch.no_grad():
        coarse = _hip(p, face, eyes, 64, dev, return_weights=True)
        assert _maxabs(coarse["w_face"], g["w_face"]) <= TOL
        zv = render.sample_zvals(pd["xy"], pd["R"], pd["T"], pd["Kinv"], n_samples=64)
        # resampling from the reference's own weights isolates FineSample parity ...
        edges = render.importance_resample(g["w_face"].to(dev), zv, n_fine=128)
        ref_edges = torch.cat([g["out_zvals"][:, 0], g["out_zvals"][:, 0, :, -1:] * 0], dim=-1)
        assert _maxabs(edges[..., :-1], g["out_zvals"][:, 0]) <= 2e-5
        fine_out = _hip(p, fine, None, 192, dev, z_edges=edges)
        assert _maxabs(fine_out["feat_face"], g["out_feat_fine"]) <= TOL
        assert _maxabs(fine_out["bg_alpha_face"], g["out_bg_alpha_fine"]) <= TOL
        # ... and the end-to-end chain from the HIP coarse weights must agree as well
        edges2 = render.importance_resample(coarse["w_face"], zv, n_fine=128)
        fine2 = _hip(p, fine, None, 192, dev, z_edges=edges2)
        assert _maxabs(fine2["feat_face"], g["out_feat_fine"]) <= 5 * TOL
    del ref_edges


def test_resample_random_u_vs_oracle():
    dev = _dev()
    g = load_golden("g1_fine")
    w, z = g["w_face"], g["zvals"]
    out = render.importance_resample(w.to(dev), z.to(dev), n_fine=int(g["n_fine"]), u=g["u"].to(dev))
    assert _maxabs(out[..., :-1], g["rnd_zvals"][:, 0]) <= 2e-5
    det = render.importance_resample(w.to(dev), z.to(dev), n_fine=int(g["n_fine"]))
    assert _maxabs(det[..., :-1], g["det_zvals"][:, 0]) <= 2e-5
    # sortedness: a size-independent property of the merged edges
    assert bool((out[..., 1:] >= out[..., :-1]).all())


def test_compositing_properties_full_size():
    """cfg2a-sized run (4096 rays x 64): size-independent properties instead of an oracle:
    weights are a sub-probability distribution, bg_alpha = 1 - sum w, rays are independent
    (a permuted ray order permutes the outputs)."""
    dev = _dev()
    p = synth.synth_problem(64, batch=1, seed=8)
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    with torch.no_grad():
        out = _hip(p, face, eyes, 64, dev, return_weights=True)
        perm = torch.randperm(4096, generator=torch.Generator().manual_seed(0))
        p2 = dict(p)
        p2["xy"] = p["xy"][:, :, perm].contiguous()
        out2 = _hip(p2, face, eyes, 64, dev)
    for tag in ("face", "eyes"):
        w = out["w_" + tag]
        assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
        assert _maxabs(1.0 - w.sum(-1), out["bg_alpha_" + tag]) <= 1e-5
        assert torch.isfinite(out["feat_" + tag]).all()
        assert torch.equal(out["feat_" + tag][:, :, perm.to(dev)], out2["feat_" + tag])


def test_errors_are_exceptions():
    dev = _dev()
    p = _to(synth.synth_problem(8), dev)
    face = _to(synth.hash_mlp_params("face"), dev)
    with pytest.raises(ValueError):
        render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                 p["appea_code"][:, :100], face, face, n_samples=32)
    from gazenerf_amd import _lib
    with pytest.raises(_lib.GnrError, match="n_samples"):
        render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                 p["appea_code"], face, face, n_samples=1000)
    with pytest.raises(ValueError, match="precision"):
        render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                 p["appea_code"], face, face, n_samples=32, precision="fp16")


# ----------------------------------------------------------------------------- backward
# Gradient tolerances.  The function is piecewise (ReLU) with sin/cos of arguments up to ~1.7e3 rad, so
# an fp32 ulp in a pre-activation near zero flips a mask and moves the gradients of a whole channel.
# Calibration (tests/diagnostics/gpu_grad_probe.py and the same script on CPU, 2 x 40 rays x 64 samples): the
# reference's own fp32 autograd differs from its fp64 run by rel-L2 up to 1.0e-2 (dR) / 6.7e-3 (trunk
# weights) and max-abs up to 1.6e-2 of scale; the HIP backward differs from fp64 by no more than that
# on every tensor (5.3e-3 on dR), and by ~1e-6 on the layers above the first ReLU mask.  A wrong kernel
# is off by O(1).  Bounds: rel-L2 <= 2e-2, max-abs <= 4e-2 of the tensor's scale.
GRAD_L2 = 2e-2
GRAD_MAX = 4e-2


def _grads_hip(p, face, eyes, n_samples, t_rand, dev, precision="fp32", hidden=384):
    pd = _to(p, dev)
    leaves = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
    ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
    out = render.render_two_stream(pd["xy"], leaves["R"], leaves["T"], pd["Kinv"], leaves["shape_code"],
                                   leaves["gaze"], leaves["appea_code"], fp, ep, n_samples=n_samples,
                                   t_rand=t_rand.to(dev) if t_rand is not None else None, precision=precision,
                                   hidden=hidden)
    loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
    loss.backward()
    return out, leaves, fp, ep


def _grads_hip_sum(p, face, eyes, n_samples, t_rand, dev, precision="fp32"):
    """As _grads_hip with a loss that is a plain sum over images and rays (no 1 / B factor)."""
    pd = _to(p, dev)
    leaves = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
    ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
    out = render.render_two_stream(pd["xy"], leaves["R"], leaves["T"], pd["Kinv"], leaves["shape_code"],
                                   leaves["gaze"], leaves["appea_code"], fp, ep, n_samples=n_samples,
                                   t_rand=t_rand.to(dev), precision=precision)
    loss = sum((out["feat_" + t] ** 2).sum() * 1e-3 + out["bg_alpha_" + t].sum() for t in ("face", "eyes"))
    loss.backward()
    return out, leaves, fp, ep


def _check_grad(name, got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = max(float(ref.abs().max()), 1e-30)
    err = float((got - ref).abs().max())
    l2 = float((got - ref).norm() / max(float(ref.norm()), 1e-30))
    assert err <= GRAD_MAX * scale and l2 <= GRAD_L2, \
        "%s: max-abs %.3e (scale %.3e), rel-L2 %.3e" % (name, err, scale, l2)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["g6_backward", "g6b_backward512"])
def test_backward_vs_reference_fixture(name, precision):
    """g6: B=2 x 32 rays x 64 samples, train-mode jitter, opaque head; gradients of the A8 loss
    captured from the reference's own autograd (big weight tensors as every 16th row).  g6b: the same at
    featmap_size=512 (cfg2b geometry)."""
    dev = _dev()
    g = load_golden(name)
    face, eyes = _weights(g)
    out, leaves, fp, ep = _grads_hip(golden_problem(g), face, eyes, int(g["n_samples"]), g["t_rand"], dev, precision)
    for tag in ("face", "eyes"):
        assert _maxabs(out["feat_" + tag], g["out_feat_" + tag]) <= TOL
        assert _maxabs(out["bg_alpha_" + tag], g["out_bg_alpha_" + tag]) <= TOL
    for k, v in leaves.items():
        _check_grad("d" + k, v.grad, g["grad_" + k])
    for tag, params in (("face", fp), ("eyes", ep)):
        for name, v in params.items():
            ref = g["gradw_%s.%s" % (tag, name)]
            got = v.grad
            if got.numel() > 4096:
                got = got.reshape(got.shape[0], -1)[::16]
            _check_grad("%s.%s" % (tag, name), got.reshape(ref.shape), ref)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n_samples,n_rays,batch,train", [(64, 40, 2, True), (32, 21, 1, False), (40, 9, 2, True)])
def test_backward_vs_oracle_live(n_samples, n_rays, batch, train, precision):
    dev = _dev()
    sub = torch.arange(n_rays) * 53 % 4096
    p = synth.synth_problem(64, batch=batch, camera="9", seed=31, ray_subset=sub)
    face = synth.hash_mlp_params("face", seed=4, density_scale=30.0)
    eyes = synth.hash_mlp_params("eyes", seed=4, density_scale=30.0)
    t_rand = synth.synth_jitter(batch, n_rays, n_samples, seed=6) if train else None
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fo = {k: v.clone().requires_grad_(True) for k, v in face.items()}
    eo = {k: v.clone().requires_grad_(True) for k, v in eyes.items()}
    ref = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                              leaves["appea_code"], fo, eo, n_samples, t_rand=t_rand)
    O.synthetic_loss(ref).backward()
    out, hl, fp, ep = _grads_hip(p, face, eyes, n_samples, t_rand, dev, precision)
    for k in leaves:
        _check_grad("d" + k, hl[k].grad, leaves[k].grad)
    for tag, hp, op in (("face", fp, fo), ("eyes", ep, eo)):
        for name in op:
            _check_grad("%s.%s" % (tag, name), hp[name].grad, op[name].grad)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_backward_is_deterministic(precision):
    """Two identical calls give bit-identical gradients (fixed-order split reductions; the reference
    trains with cudnn.deterministic=True, train.py:57)."""
    dev = _dev()
    p = synth.synth_problem(64, batch=2, camera="2", seed=1, ray_subset=torch.arange(64) * 61 % 4096)
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    t_rand = synth.synth_jitter(2, 64, 64, seed=2)
    a = _grads_hip(p, face, eyes, 64, t_rand, dev, precision)
    b = _grads_hip(p, face, eyes, 64, t_rand, dev, precision)
    for k in a[1]:
        assert torch.equal(a[1][k].grad, b[1][k].grad), k
    for x, y in ((a[2], b[2]), (a[3], b[3])):
        for k in x:
            assert torch.equal(x[k].grad, y[k].grad), k


def test_module_train_step_and_state_dict():
    """HotPathRenderer: reference parameter names, one optimiser step lowers the A8 loss."""
    from gazenerf_amd import HotPathRenderer
    dev = _dev()
    torch.manual_seed(0)
    net = HotPathRenderer().to(dev)
    keys = set(net.state_dict().keys())
    for pre in ("fg_CD_predictor_face.", "fg_CD_predictor_eyes."):
        for k in render.PARAM_ORDER:
            assert pre + k in keys
    assert net.state_dict()["fg_CD_predictor_face.FeaExt_module_5.weight"].shape == (384, 628, 1, 1)
    ref_sd = {("fg_CD_predictor_face." + k): v for k, v in synth.hash_mlp_params("face", density_scale=50.0).items()}
    ref_sd.update({("fg_CD_predictor_eyes." + k): v for k, v in synth.hash_mlp_params("eyes", density_scale=50.0).items()})
    missing, unexpected = net.load_state_dict(ref_sd, strict=False)
    assert not missing and not unexpected
    p = _to(synth.synth_problem(64, batch=2, camera="4", seed=2, ray_subset=torch.arange(128) * 29 % 4096), dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    t_rand = synth.synth_jitter(2, 128, 64, seed=3).to(dev)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = net(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["appea_code"], p["gaze"],
                  for_train=True, t_rand=t_rand)
        loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def test_cfg3_train_step_full_size_properties():
    """BASELINE cfg3 size (B=2 x 4096 rays x 64 samples, both streams, jitter): no oracle run at this
    size; instead size-independent properties of the backward: it is linear in the upstream gradient
    (loss x2 -> every gradient exactly x2: scaling by 2 is exact in fp32), deterministic, finite, and
    a ray permutation permutes nothing in the parameter gradients beyond fp32 summation order."""
    dev = _dev()
    p = synth.synth_problem(64, batch=2, camera="6", seed=12)
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    t_rand = synth.synth_jitter(2, 4096, 64, seed=12)

    def grads(scale, pp, tr):
        pd = _to(pp, dev)
        leaves = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
        fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
        ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
        out = render.render_two_stream(pd["xy"], leaves["R"], leaves["T"], pd["Kinv"], leaves["shape_code"],
                                       leaves["gaze"], leaves["appea_code"], fp, ep, n_samples=64, t_rand=tr.to(dev))
        loss = scale * sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
        loss.backward()
        return [v.grad for v in list(leaves.values()) + list(fp.values()) + list(ep.values())]

    g1 = grads(1.0, p, t_rand)
    g2 = grads(2.0, p, t_rand)
    for a, b in zip(g1, g2):
        assert torch.isfinite(a).all()
        assert torch.equal(2.0 * a, b)
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(1))
    p2 = dict(p)
    p2["xy"] = p["xy"][:, :, perm].contiguous()
    g3 = grads(1.0, p2, t_rand[:, perm].contiguous())
    for a, b in zip(g1, g3):
        scale = max(float(a.abs().max()), 1e-30)
        assert float((a - b).abs().max()) <= 1e-4 * scale


# ----------------------------------------------------------------------------- N2: feature-map merge
def test_merge_vs_reference_fixture_and_oracle_grads():
    from gazenerf_amd import merge_featmaps
    dev = _dev()
    g = load_golden("g7_merge")
    names = ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes", "bg_featmap", "gaze")
    cpu = [g[k].clone().requires_grad_(True) for k in names]
    gpu = [g[k].to(dev).clone().requires_grad_(True) for k in names]
    out = merge_featmaps(*gpu)
    for o, k in zip(out, ("out_merge_face", "out_eyes_planes", "out_merge")):
        assert _maxabs(o, g[k]) <= 1e-6
    ref = O.merge_featmaps(*cpu)
    wts = [torch.randn(ref[0].shape, generator=torch.Generator().manual_seed(i)) for i in range(3)]
    sum((r * w).sum() for r, w in zip(ref, wts)).backward()
    sum((o * w.to(dev)).sum() for o, w in zip(out, wts)).backward()
    for a, b, k in zip(gpu, cpu, names):
        scale = max(float(b.grad.abs().max()), 1e-30)
        assert _maxabs(a.grad, b.grad) <= 2e-5 * scale, k


def test_merge_end_to_end_with_render_op():
    """render_two_stream -> merge_featmaps at cfg2a size; gaze gets gradient through both ops."""
    from gazenerf_amd import merge_featmaps
    dev = _dev()
    p = _to(synth.synth_problem(64, batch=1, seed=4), dev)
    face = _to(synth.hash_mlp_params("face", density_scale=50.0), dev)
    eyes = _to(synth.hash_mlp_params("eyes", density_scale=50.0), dev)
    gaze = p["gaze"].clone().requires_grad_(True)
    bg = torch.ones(1, 258, 4096, device=dev, requires_grad=True)
    out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], gaze, p["appea_code"],
                                   face, eyes, n_samples=64)
    mf, ep, m = merge_featmaps(out["feat_face"], out["bg_alpha_face"], out["feat_eyes"], out["bg_alpha_eyes"], bg, gaze)
    assert m.shape == (1, 258, 4096) and bool((m >= mf).all()) and bool((m >= ep).all())
    (m ** 2).mean().backward()
    assert torch.isfinite(gaze.grad).all() and float(gaze.grad.abs().sum()) > 0 and torch.isfinite(bg.grad).all()


# ----------------------------------------------------------------------------- edge cases
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("n_samples,n_rays,batch", [(2, 1, 1), (512, 3, 1), (33, 5, 2), (64, 1, 4)])
def test_edge_sizes_forward_and_backward(n_samples, n_rays, batch, precision):
    """Minimum (2) and maximum (512) sample counts, a single ray, one sample past a chunk boundary
    (33), more images than rays: forward vs the oracle on its own edges, gradients finite and equal
    to the oracle's within the fp32 noise bound."""
    dev = _dev()
    sub = (torch.arange(n_rays) * 911 + 17) % 4096
    p = synth.synth_problem(64, batch=batch, camera="11", seed=77, ray_subset=sub)
    face = synth.hash_mlp_params("face", seed=6, density_scale=10.0)
    eyes = synth.hash_mlp_params("eyes", seed=6, density_scale=10.0)
    t_rand = synth.synth_jitter(batch, n_rays, n_samples, seed=8)
    edges = O.sample_edges(p["xy"], p["R"], p["T"], p["Kinv"], n_samples, t_rand=t_rand)[0]
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fo = {k: v.clone().requires_grad_(True) for k, v in face.items()}
    eo = {k: v.clone().requires_grad_(True) for k, v in eyes.items()}
    ref = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                              leaves["appea_code"], fo, eo, n_samples, t_rand=t_rand)
    O.synthetic_loss(ref).backward()
    with torch.no_grad():
        out_e = _hip(p, face, eyes, n_samples, dev, z_edges=edges, return_weights=True, precision=precision)
    for tag in ("face", "eyes"):
        assert _maxabs(out_e["feat_" + tag], ref["feat_" + tag]) <= TOL
        assert _maxabs(out_e["bg_alpha_" + tag], ref["bg_alpha_" + tag]) <= TOL
        assert _maxabs(out_e["w_" + tag], ref["w_" + tag]) <= W_TOL
    out, hl, fp, ep = _grads_hip(p, face, eyes, n_samples, t_rand, dev, precision)
    for k in leaves:
        assert torch.isfinite(hl[k].grad).all()
    if (n_samples & (n_samples - 1)) == 0:          # power-of-two counts: same z as the oracle's sweep
        for tag, hp, op in (("face", fp, fo), ("eyes", ep, eo)):
            for name in ("RGB_layer_2.weight", "RGB_layer_0.weight", "density_module.weight", "FeaExt_module_7.bias"):
                _check_grad("%s.%s" % (tag, name), hp[name].grad, op[name].grad)


def test_more_images_than_workgroup_slots():
    """144 images x 16 rays x 32 samples in ONE call against the same images as two calls of 72: per-image gradients equal, weight
    gradients = the sum of the halves'.  With more images than one round of workgroups a weight-gradient GEMM runs one split per
    image, and until round 5 its rider shares (bias column sums, the density-head dot) outgrew their fixed scratch from 65 images
    on -- wrong `density_module.weight` / bias gradients, from 129 images on also overwritten partial tiles."""
    dev = _dev()
    B, n_rays, n_samples = 144, 16, 32
    sub = (torch.arange(n_rays) * 251 + 5) % 4096
    p = synth.synth_problem(64, batch=B, camera="3", seed=21, ray_subset=sub)
    face = synth.hash_mlp_params("face", seed=6, density_scale=10.0)
    eyes = synth.hash_mlp_params("eyes", seed=6, density_scale=10.0)
    t_rand = synth.synth_jitter(B, n_rays, n_samples, seed=9)

    def run(lo, hi):
        q = {k: v[lo:hi].contiguous() for k, v in p.items()}
        out, leaves, fp, ep = _grads_hip_sum(q, face, eyes, n_samples, t_rand[lo:hi], dev)
        return {k: v.grad.double().cpu() for k, v in leaves.items()}, {k: v.grad.double().cpu() for k, v in fp.items()}, \
            {k: v.grad.double().cpu() for k, v in ep.items()}

    full = run(0, B)
    a, b = run(0, B // 2), run(B // 2, B)
    for k in full[0]:                                  # per-image inputs: the halves side by side
        ref = torch.cat([a[0][k], b[0][k]], 0)
        assert float((full[0][k] - ref).norm() / ref.norm().clamp_min(1e-30)) <= 1e-5, k
    for tag, i in (("face", 1), ("eyes", 2)):
        for k in full[i]:
            ref = a[i][k] + b[i][k]
            assert float((full[i][k] - ref).norm() / ref.norm().clamp_min(1e-30)) <= 2e-5, "%s.%s" % (tag, k)


def test_empty_and_invalid_inputs_are_rejected():
    """The reference would fail inside PyTorch on empty tensors; the C ABI rejects them up front."""
    dev = _dev()
    from gazenerf_amd import _lib
    p = _to(synth.synth_problem(8), dev)
    face = _to(synth.hash_mlp_params("face"), dev)
    empty_xy = p["xy"][:, :, :0].contiguous()
    with pytest.raises(_lib.GnrError, match="empty problem"):
        render.render_two_stream(empty_xy, p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                 face, face, n_samples=32)
    with pytest.raises(_lib.GnrError, match="hidden=384"):
        render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                 _to(synth.hash_mlp_params("face", hidden=33), dev), None, n_samples=32, hidden=33)
    with pytest.raises(TypeError):
        render.render_two_stream(p["xy"].double(), p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                 p["appea_code"], face, face, n_samples=32)


# ----------------------------------------------------------------------------- cfg2b at full size (512 x 512 rays)
def test_cfg2b_full_size_forward_properties():
    """BASELINE cfg2b at its real size (featmap_size=512: 262 144 rays x 64 samples, both streams): no oracle
    run at this size; size-independent properties instead.  Finite; bg_alpha = 1 - sum w; rays are independent:
    a ray permutation permutes the outputs bit for bit, and the result does not depend on how the rays are split
    over calls (micro-batches)."""
    dev = _dev()
    n = 512 * 512
    p = _to(synth.synth_problem(512, batch=1, camera="3", seed=100), dev)
    face = _to(synth.hash_mlp_params("face", seed=0, density_scale=50.0), dev)
    eyes = _to(synth.hash_mlp_params("eyes", seed=0, density_scale=50.0), dev)
    call = lambda xy, **kw: render.render_two_stream(xy, p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                                     p["appea_code"], face, eyes, n_samples=64, **kw)
    with torch.no_grad():
        out = call(p["xy"], return_weights=True)
        for tag in ("face", "eyes"):
            w = out["w_" + tag]
            assert torch.isfinite(out["feat_" + tag]).all() and torch.isfinite(out["bg_alpha_" + tag]).all()
            assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-5
            assert _maxabs(1.0 - w.sum(-1), out["bg_alpha_" + tag]) <= 1e-5
        del w
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).to(dev)
        out_p = call(p["xy"][:, :, perm].contiguous())
        for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes"):
            assert torch.equal(out[k][:, :, perm], out_p[k]), k
        del out_p
        # split into three unequal micro-batches (sizes that are not multiples of the workgroup's 4 chunks)
        cuts = [0, 100003, 100003 + 37, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = call(p["xy"][:, :, a:b].contiguous())
            for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes"):
                assert torch.equal(out[k][:, :, a:b], part[k]), (k, a, b)
    # The fixture rays (the reference's own outputs at side 512: oracle/gen_golden_s512.py) are a subset of this very
    # pixel grid: render the FULL 512 x 512 image with each fixture's codes / camera / weights and compare its 256 rays
    # inside it, both precisions -- the bench's own image size against reference outputs, not only against itself.
    for name in ("g2b_np64_opaque", "g2b_np64_frontal", "g2b_np64_orbit3"):
        g = load_golden(name)
        sub = g["ray_subset"].to(dev)
        assert torch.equal(p["xy"][:, :, sub].cpu(), g["in_xy"])
        q = _to(golden_problem(g), dev)
        gface, geyes = _weights(g)
        gface, geyes = _to(gface, dev), _to(geyes, dev)
        for precision in PRECISIONS:
            with torch.no_grad():
                full = render.render_two_stream(p["xy"], q["R"], q["T"], q["Kinv"], q["shape_code"], q["gaze"], q["appea_code"],
                                                gface, geyes, n_samples=64, precision=precision)
            for tag in ("face", "eyes"):
                assert _maxabs(full["feat_" + tag][:, :, sub], g["out_feat_" + tag]) <= TOL, (name, precision, tag)
                assert _maxabs(full["bg_alpha_" + tag][:, :, sub], g["out_bg_alpha_" + tag]) <= TOL, (name, precision, tag)
            del full


@pytest.mark.parametrize("precision", PRECISIONS)
def test_cfg2b_full_size_backward_is_independent_of_ray_tiling(precision):
    """A whole 512 x 512-ray image trains through ONE call of the op (the reference makes one forward call per
    image, models/gaze_nerf.py:211-320): the op tiles the rays internally within its workspace budget (540 GB of
    saved activations would be needed otherwise).  Gradients must not depend on the tile size beyond fp32
    summation order, and the forward outputs not at all."""
    dev = _dev()
    p = _to(synth.synth_problem(512, batch=1, camera="3", seed=100), dev)
    t_rand = synth.synth_jitter(1, 512 * 512, 64, seed=7).to(dev)
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)

    def grads(**kw):
        leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
        fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
        ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
        out = render.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                                       leaves["appea_code"], fp, ep, n_samples=64, t_rand=t_rand, precision=precision, **kw)
        loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
        loss.backward()
        outs = {k: out[k].detach() for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes")}
        return outs, [v.grad for v in list(leaves.values()) + list(fp.values()) + list(ep.values())]

    o1, g1 = grads()                       # default budget -> the op picks its own tile
    o2, g2 = grads(ray_tile=16384 + 256)   # a different, non-dividing tile
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    for a, b in zip(g1, g2):
        assert torch.isfinite(a).all()
        scale = max(float(a.abs().max()), 1e-30)
        assert float((a - b).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("precision", PRECISIONS)
def test_ray_tiling_matches_untiled(precision):
    """In-op ray tiling (ragged last tile) against the single-shot path on a problem small enough for both."""
    dev = _dev()
    n_rays = 600
    p = _to(synth.synth_problem(64, batch=2, camera="4", seed=3, ray_subset=torch.arange(n_rays) * 13 % 4096), dev)
    t_rand = synth.synth_jitter(2, n_rays, 64, seed=9).to(dev)
    face = synth.hash_mlp_params("face", seed=1, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=1, density_scale=50.0)

    def grads(**kw):
        leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
        fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
        ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
        out = render.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                                       leaves["appea_code"], fp, ep, n_samples=64, t_rand=t_rand, precision=precision,
                                       return_depth=True, **kw)
        loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
        loss.backward()
        return {k: v.detach() for k, v in out.items()}, [v.grad for v in list(leaves.values()) + list(fp.values()) + list(ep.values())]

    o1, g1 = grads()
    o2, g2 = grads(ray_tile=256)
    o3, g3 = grads(ws_budget_bytes=1 << 28)          # budget-driven tile choice
    for k in o1:
        # the training forward (saving kernel) and the inference forward of the tiled mode share every instruction
        # of the arithmetic chain
        assert _maxabs(o1[k], o2[k]) <= 1e-6 and torch.equal(o2[k], o3[k]), k
    for a, b, c in zip(g1, g2, g3):
        scale = max(float(a.abs().max()), 1e-30)
        assert float((a - b).abs().max()) <= 1e-4 * scale
        assert float((a - c).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize("precision", PRECISIONS)
def test_tiled_training_entry_matches_the_single_call(precision):
    """render_two_stream_tiled (round 4): per ray tile forward-with-save -> the caller's per-ray loss -> backward, 3/3 of
    the FLOPs.  Against ONE render_two_stream call + backward on a problem small enough for it: identical outputs (the
    same saving kernel runs on every tile), gradients equal to fp32 summation order (tiles are summed one after the
    other), the total loss, a ragged last tile, and bit-identical gradients when repeated."""
    dev = _dev()
    n_rays, B, C = 600, 2, 258
    p = _to(synth.synth_problem(64, batch=B, camera="4", seed=3, ray_subset=torch.arange(n_rays) * 13 % 4096), dev)
    t_rand = synth.synth_jitter(B, n_rays, 64, seed=9).to(dev)
    face = synth.hash_mlp_params("face", seed=1, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=1, density_scale=50.0)
    target = torch.linspace(-1.0, 1.0, B * C * n_rays, device=dev).reshape(B, C, n_rays)     # an image-like target per ray

    def share(out, sl):          # this tile's share of mean((feat - target)^2) + mean(bg_alpha), both streams
        n = sl.stop - sl.start
        return sum(((out["feat_" + t] - target[:, :, sl]) ** 2).sum() / (B * C * n_rays) + out["bg_alpha_" + t].sum() / (B * n_rays)
                   for t in ("face", "eyes")) + 0.0 * n

    def leaves_():
        lv = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
        fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
        ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
        return lv, fp, ep

    lv, fp, ep = leaves_()
    out = render.render_two_stream(p["xy"], lv["R"], lv["T"], p["Kinv"], lv["shape_code"], lv["gaze"], lv["appea_code"], fp, ep,
                                   n_samples=64, t_rand=t_rand, precision=precision)
    loss1 = share(out, slice(0, n_rays))
    loss1.backward()
    g1 = [v.grad for v in list(lv.values()) + list(fp.values()) + list(ep.values())]

    def tiled(tile):
        lv, fp, ep = leaves_()
        total, outs = render.render_two_stream_tiled(p["xy"], lv["R"], lv["T"], p["Kinv"], lv["shape_code"], lv["gaze"],
                                                     lv["appea_code"], fp, ep, loss_fn=share, n_samples=64, ray_tile=tile,
                                                     t_rand=t_rand, precision=precision, return_outputs=True)
        return total, outs, [v.grad for v in list(lv.values()) + list(fp.values()) + list(ep.values())]

    total, outs, g2 = tiled(256)                 # 256 + 256 + 88 rays
    assert abs(float(total) - float(loss1)) <= 1e-5 * abs(float(loss1))
    for k in out:
        assert torch.equal(outs[k], out[k].detach()), k
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-4 * max(float(a.abs().max()), 1e-30)
    _, _, g3 = tiled(256)
    for a, b in zip(g2, g3):
        assert torch.equal(a, b)
    total4, _, g4 = tiled(None)                  # budget-driven: everything fits -> one tile == the single call
    for a, b in zip(g1, g4):
        assert torch.equal(a, b)
    with pytest.raises(ValueError, match="scalar"):
        render.render_two_stream_tiled(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                       {k: v.to(dev).requires_grad_(True) for k, v in face.items()}, None,
                                       loss_fn=lambda o, sl: o["feat_face"], n_samples=64, ray_tile=256)


def test_save_for_backward_guards():
    """An in-place parameter update between forward and backward is an error (autograd's version check), and so is
    a second backward without retain_graph -- not silently inconsistent gradients."""
    dev = _dev()
    p = _to(synth.synth_problem(64, batch=1, seed=2, ray_subset=torch.arange(32)), dev)
    face = {k: v.to(dev).requires_grad_(True) for k, v in synth.hash_mlp_params("face").items()}
    out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                   face, None, n_samples=32)
    with torch.no_grad():
        face["FeaExt_module_3.weight"].mul_(0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out["feat_face"].sum().backward()
    out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                   face, None, n_samples=32)
    loss = out["feat_face"].sum()
    loss.backward(retain_graph=True)
    g1 = face["RGB_layer_2.bias"].grad.clone()
    face["RGB_layer_2.bias"].grad = None
    loss.backward()                                   # retained graph: a second backward works and agrees
    assert torch.equal(g1, face["RGB_layer_2.bias"].grad)
    with pytest.raises(RuntimeError):
        loss.backward()


# ----------------------------------------------------------------------------- backward: noise-relative gates
def _all_grads(pp, f, e, tr, fn, loss_fn=None):
    leaves = {k: pp[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.clone().requires_grad_(True) for k, v in f.items()}
    ep = {k: v.clone().requires_grad_(True) for k, v in e.items()}
    out = fn(pp["xy"], leaves["R"], leaves["T"], pp["Kinv"], leaves["shape_code"], leaves["gaze"], leaves["appea_code"],
             fp, ep, tr)
    loss = sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
    loss.backward()
    g = {"d" + k: v.grad for k, v in leaves.items()}
    for tag, d in (("face", fp), ("eyes", ep)):
        for k, v in d.items():
            g[tag + "." + k] = v.grad
    return g


def _grad_problem(stable: bool, draw: int = 0):
    """stable=False: the opaque-head / train-jitter case of the other backward tests (ReLU masks flip under fp32
    noise: the reference's own fp32 autograd is ~1e-2 rel-L2 from its fp64 run on dR).
    stable=True: biases x100 keep every pre-activation far from 0 (no mask can flip), a small positive density keeps
    both streams translucent: there the reference's fp32 noise is <= 7e-5 rel-L2 on every parameter tensor and
    ~2e-3 on dR / dT (fp32 sin/cos arguments up to 1.7e3 rad), so a dropped 1 % term is visible."""
    n_rays, B = 24, 2
    # draw > 0: another camera, other rays, codes, jitter and weights -- an independent draw of the same distribution
    p = synth.synth_problem(64, batch=B, camera=str((9 + 5 * draw) % 45), seed=31 + draw,
                            ray_subset=(torch.arange(n_rays) * 53 + 977 * draw) % 4096)
    t_rand = synth.synth_jitter(B, n_rays, 64, seed=6 + draw)
    if stable:
        face = synth.hash_mlp_params("face", seed=4 + draw, density_scale=0.02, bias_scale=100.0)
        eyes = synth.hash_mlp_params("eyes", seed=4 + draw, density_scale=0.02, bias_scale=100.0)
        face["density_module.bias"] += 0.3
        eyes["density_module.bias"] += 0.3
    else:
        face = synth.hash_mlp_params("face", seed=4 + draw, density_scale=30.0)
        eyes = synth.hash_mlp_params("eyes", seed=4 + draw, density_scale=30.0)
    return p, face, eyes, t_rand


# floor for tensors where the reference's fp32 noise is ~1e-6 (the layers above the last ReLU mask): the fp32 kernels
# sum in another order (~1e-6); a 3-term bf16 split keeps ~16 mantissa bits per operand, i.e. ~6e-5 on a gradient
GRAD_EPS = {"fp32": 2e-5, "bf16x3": 1.5e-4}
GRAD_STABLE_ABS = {"fp32": 2e-4, "bf16x3": 4e-4}   # mask-stable problem: bound on every parameter / latent gradient


def _permuted_network(params, seed):
    """The SAME function with every hidden layer's channels permuted (rows of W_l and b_l, the matching input columns of the
    layers that read them): mathematically identical, but fp32 sums every dot product in another order -- an independent
    draw of the reference's own fp32 rounding noise.  Returns (permuted params, un-permute for a gradient dict)."""
    g = torch.Generator().manual_seed(seed)
    H, vp = 384, 244
    P = {i: torch.randperm(H, generator=g) for i in range(8)}
    Pr0, Pr1 = torch.randperm(H, generator=g), torch.randperm(H // 2, generator=g)
    ident = lambda n: torch.arange(n)
    rows, cols = {}, {}
    for i in range(8):
        name = "FeaExt_module_%d" % i
        rows[name] = P[i]
        cols[name] = ident(vp) if i == 0 else (torch.cat([ident(vp), vp + P[4]]) if i == 5 else P[i - 1])
    rows["density_module"], cols["density_module"] = ident(1), P[7]
    rows["RGB_layer_0"], cols["RGB_layer_0"] = Pr0, P[7]
    rows["RGB_layer_1"], cols["RGB_layer_1"] = Pr1, torch.cat([Pr0, H + ident(params["RGB_layer_1.weight"].shape[1] - H)])
    rows["RGB_layer_2"], cols["RGB_layer_2"] = ident(params["RGB_layer_2.weight"].shape[0]), Pr1
    out = {}
    for name in rows:
        w = params[name + ".weight"]
        w2 = w.reshape(w.shape[0], w.shape[1])[rows[name]][:, cols[name]]
        out[name + ".weight"] = w2.reshape(w.shape).contiguous()
        out[name + ".bias"] = params[name + ".bias"][rows[name]].contiguous()

    def unpermute(name, grad):          # name: "FeaExt_module_3.weight" ...
        base, kind = name.rsplit(".", 1)
        if kind == "bias":
            res = torch.empty_like(grad)
            res[rows[base]] = grad
            return res
        g2 = grad.reshape(grad.shape[0], grad.shape[1])
        res = torch.empty_like(g2)
        res[rows[base][:, None], cols[base][None, :]] = g2
        return res.reshape(grad.shape)
    return out, unpermute


def _noise_errors(stable, draw, precisions, dev, with_null=False):
    """Per tensor: rel-L2 error against the fp64 oracle of (the oracle's own fp32 autograd, the HIP gradient per precision)."""
    p, face, eyes, t_rand = _grad_problem(stable, draw)
    d64 = lambda d: {k: v.double() for k, v in d.items()}
    ofn = lambda xy, R, T, K, s, g, a, f, e, tr: O.render_two_stream(xy, R, T, K, s, g, a, f, e, 64, t_rand=tr)
    exact = _all_grads(d64(p), d64(face), d64(eyes), t_rand.double(), ofn)
    ref32 = _all_grads(p, face, eyes, t_rand, ofn)
    hips = {pr: _all_grads(_to(p, dev), _to(face, dev), _to(eyes, dev), t_rand.to(dev),
                           lambda xy, R, T, K, s, g, a, f, e, tr, pr=pr: render.render_two_stream(
                               xy, R, T, K, s, g, a, f, e, n_samples=64, t_rand=tr, precision=pr)) for pr in precisions}
    null = None
    if with_null:
        # the reference's OWN fp32 arithmetic on a channel-permuted copy of the network: a second, independent draw of its
        # rounding noise -- the yardstick for "another fp32 arithmetic" that involves no HIP code at all
        fperm, fun = _permuted_network(face, 1000 + draw)
        eperm, eun = _permuted_network(eyes, 2000 + draw)
        raw = _all_grads(p, fperm, eperm, t_rand, ofn)
        null = {}
        for k, v in raw.items():
            null[k] = fun(k[5:], v) if k.startswith("face.") else (eun(k[5:], v) if k.startswith("eyes.") else v)
    out = {}
    for k, r in exact.items():
        n = max(float(r.norm()), 1e-30)
        e_hip = {pr: float((h[k].cpu().double() - r).norm()) / n for pr, h in hips.items()}
        if null is not None:
            e_hip["null"] = float((null[k].double() - r).norm()) / n
        out[k] = (float((ref32[k].double() - r).norm()) / n, e_hip)
    return out


@pytest.mark.parametrize("precision", PRECISIONS)
def test_backward_error_is_within_the_reference_fp32_noise_mask_stable(precision):
    """Per tensor (all 53): rel-L2 error of the HIP gradient against the oracle in fp64 <= 1.25 x the error of the oracle's
    own fp32 autograd against fp64 + eps -- the forward's criterion (test_bf16x3_is_as_close_to_exact_as_the_reference_fp32)
    applied to the backward, on the problem where no ReLU mask can flip: there it is a bound of <= 2e-4 (fp32) / 4e-4
    (bf16x3) rel-L2 on every parameter / latent gradient."""
    errs = _noise_errors(True, 0, [precision], _dev())
    eps = GRAD_EPS[precision]
    bad = []
    for k, (e_ref, e_hip) in errs.items():
        if not e_hip[precision] <= 1.25 * e_ref + eps:
            bad.append("%s: hip %.2e ref32 %.2e" % (k, e_hip[precision], e_ref))
        if k not in ("dR", "dT"):
            assert e_hip[precision] <= GRAD_STABLE_ABS[precision], (k, e_hip[precision])
    assert not bad, bad


N_NOISE_DRAWS = 16
# per tensor over the draws: median of err_hip / err_ref32 (the review's gate), and the tail.
# fp32 kernels: per-tensor max against a yardstick that involves no HIP code + share of all (tensor, draw) ratios above 3
# (measured, profiles/r4_grad_noise_draws.txt: the reference's OWN second fp32 draw -- channel-permuted network -- reaches 4.3
# on the density head; the fp32 kernels 3.75, 0.5 % of the ratios above 3).
# bf16x3 (round 5): its 3-term products carry ~16 mantissa bits, so more pre-activations sit within
# its noise of zero and ONE flipped mask moves a whole layer pair's tensors by a factor of ~12 in that draw (observed once in
# 8 draws).  Round 4 let that through with "max <= 20, 5 % of the ratios above 3", which would also pass a kernel that
# occasionally drops a layer's contribution on one tensor.  Now, per tensor over 16 draws: AT MOST ONE draw above
# `outlier` = 5 (or above 1.25 x the yardstick's own worst ratio on that tensor, where the yardstick itself exceeds 5), and that one
# draw bounded in ABSOLUTE terms: rel-L2 against fp64 <= `outlier_abs` = 1e-2, the reference's own fp32 noise on its noisiest
# tensors (dR).  A ratio cap cannot do that job: the 16 draws contain one (draw 13, eyes stream) in which the reference's fp32 run
# flips no mask at all -- its noise on eyes.FeaExt_module_0..4 is 1e-5 -- while the bf16x3 kernels flip ONE (3e-3 on those
# tensors, ratio 90-215; tests/diagnostics/gpu_x3_outlier.py shows the single channel / single row it sits in,
# profiles/r5_x3_outlier_draw13.txt).  A kernel that drops a layer's contribution on one tensor is off by O(1), not 3e-3.
# profiles/r5_grad_noise_draws.txt is the distribution;
# profiles/r5_grad_gate_dropped_term.txt shows the gate failing when ONE cross term (W_lo x a_hi) is dropped from ONE layer of
# bwd3_chain_kernel (an experimental build, -DGNR_ABLATE=128: a diagnostic, not CI).
NOISE_GATE = {"fp32": dict(median=1.25, max_floor=3.0, max_vs_null=1.25, share_above_3=0.02),
              "bf16x3": dict(median=1.5, outlier=5.0, outlier_vs_null=1.25, outliers_allowed=1, outlier_abs=1e-2)}


def noise_gate_failures(ratios, errs, precisions=None):
    """ratios[who][tensor] = [err_who / err_ref32 per draw], errs[who][tensor] = [err_who per draw] (rel-L2 against the fp64
    oracle), who in PRECISIONS + ["null"] -> list of violated gates."""
    import statistics
    null_worst = max(max(rs) for rs in ratios["null"].values())
    bad = []
    for pr in (precisions or PRECISIONS):
        gate = NOISE_GATE[pr]
        for k, rs in ratios[pr].items():
            med, mx = statistics.median(rs), max(rs)
            line = "%s %s: median %.2f max %.2f (%s)" % (pr, k, med, mx, " ".join("%.2f" % r for r in rs))
            if med > gate["median"]:
                bad.append(line + " -- median gate %.2f" % gate["median"])
            if "outlier" in gate:
                thr = max(gate["outlier"], gate["outlier_vs_null"] * max(ratios["null"][k]))
                above = [i for i, r in enumerate(rs) if r > thr]
                worst_abs = max([errs[pr][k][i] for i in above], default=0.0)
                if len(above) > gate["outliers_allowed"] or worst_abs > gate["outlier_abs"]:
                    bad.append(line + " -- %d draw(s) above %.2f (allowed %d), their worst rel-L2 %.2e (allowed %.0e)" % (
                        len(above), thr, gate["outliers_allowed"], worst_abs, gate["outlier_abs"]))
            else:
                max_gate = max(gate["max_floor"], gate["max_vs_null"] * null_worst)
                if mx > max_gate:
                    bad.append(line + " -- max gate %.2f" % max_gate)
        if "share_above_3" in gate:
            allr = [r for rs in ratios[pr].values() for r in rs]
            share = sum(r > 3.0 for r in allr) / len(allr)
            if share > gate["share_above_3"]:
                bad.append("%s: %.3f of all ratios above 3 (gate %.2f)" % (pr, share, gate["share_above_3"]))
    return bad


def noise_ratios(n_draws, dev):
    """-> (ratios, errs): per arithmetic and tensor the list over the draws of err / err_ref32 and of err itself."""
    ratios = {pr: {} for pr in PRECISIONS + ["null"]}
    errs = {pr: {} for pr in PRECISIONS + ["null"]}
    for draw in range(n_draws):
        for k, (e_ref, e_hip) in _noise_errors(False, draw, PRECISIONS, dev, with_null=True).items():
            for pr in ratios:
                # eps: the floor where the reference's own noise is ~1e-6 (the layers above the last ReLU mask)
                eps = GRAD_EPS.get(pr, GRAD_EPS["fp32"])
                ratios[pr].setdefault(k, []).append(max(e_hip[pr] - eps, 0.0) / max(e_ref, 1e-30))
                errs[pr].setdefault(k, []).append(e_hip[pr])
    return ratios, errs


def test_backward_error_follows_the_reference_fp32_noise_distribution():
    """The unstable problem (opaque head + train jitter: ReLU masks flip under fp32 noise).  Where masks flip the error IS
    the set of flipped samples -- a discrete draw per arithmetic, and with 2 x 24 rays one flip moves a tensor's error by a
    factor of a few.  Round 3 asserted "same distribution, another draw" with ONE draw and a loosened factor (2.5); this
    test measures it.  16 independent problems (cameras, rays, codes, jitter, weights); per tensor (all 53) the ratio
    err_hip / err_ref32, both against the oracle in fp64:
      * median over the draws <= 1.25 (fp32 kernels) / 1.5 (bf16x3)  -- observed worst 1.13 / 0.94: the kernels are, if
        anything, closer to fp64 than the reference's fp32 autograd.  A backward that drops a 1 % term is off by 10-2000 x the
        noise in EVERY draw and fails here;
      * the tail against a yardstick that involves no HIP code: the reference's own fp32 autograd on a channel-permuted
        copy of the network (same function, other summation order = its own second draw).  fp32 kernels: per-tensor max
        <= 1.25 x the yardstick's worst ratio (floor 3) and at most 2 % of all (tensor, draw) ratios above 3; bf16x3: per
        tensor at most one draw above 5, and that one within 1e-2 rel-L2 of fp64 (see NOISE_GATE)."""
    bad = noise_gate_failures(*noise_ratios(N_NOISE_DRAWS, _dev()))
    assert not bad, bad


@pytest.mark.parametrize("precision", PRECISIONS)
def test_camera_gradient_matches_finite_differences_of_the_fp64_oracle(precision):
    """dL/dR and dL/dT of the HIP backward against central finite differences of the fp64 oracle along random
    directions (mask-stable problem; fp64 keeps the difference quotient clean despite sin(512 x))."""
    dev = _dev()
    p, face, eyes, t_rand = _grad_problem(True)
    d64 = lambda d: {k: v.double() for k, v in d.items()}
    p64, f64, e64, tr64 = d64(p), d64(face), d64(eyes), t_rand.double()

    def loss_at(R, T):
        with torch.no_grad():
            out = O.render_two_stream(p64["xy"], R, T, p64["Kinv"], p64["shape_code"], p64["gaze"], p64["appea_code"],
                                      f64, e64, 64, t_rand=tr64)
            return float(O.synthetic_loss(out))

    hip = _all_grads(_to(p, dev), _to(face, dev), _to(eyes, dev), t_rand.to(dev),
                     lambda xy, R, T, K, s, g, a, f, e, tr: render.render_two_stream(
                         xy, R, T, K, s, g, a, f, e, n_samples=64, t_rand=tr, precision=precision))
    gR, gT = hip["dR"].cpu().double(), hip["dT"].cpu().double()
    gen = torch.Generator().manual_seed(5)
    h = 1e-6
    for trial in range(4):
        dR = torch.randn(p64["R"].shape, generator=gen, dtype=torch.float64) * (trial != 1)
        dT = torch.randn(p64["T"].shape, generator=gen, dtype=torch.float64) * (trial != 2)
        fd = (loss_at(p64["R"] + h * dR, p64["T"] + h * dT) - loss_at(p64["R"] - h * dR, p64["T"] - h * dT)) / (2 * h)
        an = float((gR * dR).sum() + (gT * dT).sum())
        bound = 1e-2 * float(torch.sqrt((gR ** 2).sum() + (gT ** 2).sum()) * torch.sqrt((dR ** 2).sum() + (dT ** 2).sum()))
        assert abs(fd - an) <= bound, (trial, fd, an, bound)


def test_hierarchical_camera_gradient_follows_the_reference():
    """FineSample detaches only the weights (utils/model_utils.py:418): the merged fine edges still shift 1:1 with
    T_z through the coarse z they interpolate (:455-476), so dL/dT of the fine pass is the plane sweep's
    (m_x A_0 + m_y A_1), not that of constant edges (A_2).  HotPathRenderer(hier_sampling=True) against the
    oracle's composition of the same sub-modules, loss on the fine outputs only."""
    from gazenerf_amd import HotPathRenderer
    dev = _dev()
    n_rays = 24
    p = synth.synth_problem(64, batch=2, camera="7", seed=41, ray_subset=torch.arange(n_rays) * 97 % 4096)
    face = synth.hash_mlp_params("face", seed=3, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=3, density_scale=50.0)
    fine = synth.hash_mlp_params("fine", seed=3, density_scale=50.0)
    R = p["R"].clone().requires_grad_(True)
    T = p["T"].clone().requires_grad_(True)
    coarse = O.render_two_stream(p["xy"], R, T, p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes, 64)
    of = O.hier_fine_pass(coarse, p["shape_code"], p["gaze"], p["appea_code"], fine, 128)
    ((of["feat_fine"] ** 2).mean() + of["bg_alpha_fine"].mean()).backward()

    net = HotPathRenderer(hier_sampling=True).to(dev)
    sd = {}
    for pre, params in (("fg_CD_predictor_face.", face), ("fg_CD_predictor_eyes.", eyes), ("fine_fg_CD_predictor.", fine)):
        sd.update({pre + k: v for k, v in params.items()})
    net.load_state_dict(sd, strict=True)
    pd = _to(p, dev)
    Rg = pd["R"].clone().requires_grad_(True)
    Tg = pd["T"].clone().requires_grad_(True)
    out = net(pd["xy"], Rg, Tg, pd["Kinv"], pd["shape_code"], pd["appea_code"], pd["gaze"])
    assert _maxabs(out["feat_fine"], of["feat_fine"]) <= 5 * TOL
    ((out["feat_fine"] ** 2).mean() + out["bg_alpha_fine"].mean()).backward()
    _check_grad("hier dT", Tg.grad, T.grad)
    _check_grad("hier dR", Rg.grad, R.grad)
    # the constant-edges formula differs at O(1) in dT_z: make sure the test can tell them apart
    edges = out["fine_edges"].detach()
    Tc = pd["T"].clone().requires_grad_(True)
    fo = render.render_two_stream(pd["xy"], pd["R"], Tc, pd["Kinv"], pd["shape_code"], pd["gaze"], pd["appea_code"],
                                  _to(fine, dev), None, n_samples=192, z_edges=edges)
    ((fo["feat_face"] ** 2).mean() + fo["bg_alpha_face"].mean()).backward()
    assert float((Tc.grad.cpu() - T.grad).abs().max()) > 0.2 * float(T.grad.abs().max())


# ----------------------------------------------------------------------------- the two bindings of the C ABI
@pytest.mark.parametrize("precision", PRECISIONS)
def test_torch_extension_and_ctypes_bindings_agree(precision, monkeypatch):
    """The PyTorch C++ extension (default when built) and the ctypes binding call the same C ABI: bit-identical
    outputs and gradients; and the extension really is the active binding on this box."""
    from gazenerf_amd import _torch_ext
    dev = _dev()
    assert _torch_ext.load(required=True) is not None
    p = synth.synth_problem(64, batch=2, camera="2", seed=1, ray_subset=torch.arange(80) * 61 % 4096)
    face = synth.hash_mlp_params("face", seed=0, density_scale=50.0)
    eyes = synth.hash_mlp_params("eyes", seed=0, density_scale=50.0)
    t_rand = synth.synth_jitter(2, 80, 64, seed=2)
    res = {}
    for binding in ("torch_ext", "ctypes"):
        monkeypatch.setenv("GNR_BINDING", binding)
        assert (_torch_ext.active() is not None) == (binding == "torch_ext")
        res[binding] = _grads_hip(p, face, eyes, 64, t_rand, dev, precision)
    a, b = res["torch_ext"], res["ctypes"]
    for k in ("feat_face", "bg_alpha_face", "feat_eyes", "bg_alpha_eyes"):
        assert torch.equal(a[0][k], b[0][k]), k
    for i in (1, 2, 3):
        for k in a[i]:
            assert torch.equal(a[i][k].grad, b[i][k].grad), k
    monkeypatch.setenv("GNR_BINDING", "torch_ext")
    pd = _to(p, dev)
    with pytest.raises(RuntimeError, match="n_samples"):          # gnr_last_error() through TORCH_CHECK
        render.render_two_stream(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["gaze"], pd["appea_code"],
                                 _to(face, dev), None, n_samples=1000)


def test_cfg5_full_size_hierarchical_properties():
    """BASELINE cfg5 at its real size (512 x 512 rays: coarse 64 -> FineSample -> 192-sample fine pass through a third
    MLP): size-independent properties -- merged edges sorted and inside the coarse span, every coarse edge present in
    the merged row, finite fine outputs, bg_alpha = 1 - sum w, and the fixture rays (reference outputs, g5b) reproduced
    when the same rays are rendered as part of the full image."""
    dev = _dev()
    g = load_golden("g5b_hier512")
    p = golden_problem(g)                                    # frontal camera, the fixture's latent codes
    full = dict(p)
    full["xy"] = synth.pixel_grid(512)
    full = _to(full, dev)
    face, eyes = _weights(g)
    fine = synth.hash_mlp_params("fine", seed=int(g["weight_seed"]), density_scale=float(g["density_scale"]))
    face, eyes, fine = _to(face, dev), _to(eyes, dev), _to(fine, dev)
    args = (full["R"], full["T"], full["Kinv"], full["shape_code"], full["gaze"], full["appea_code"])
    with torch.no_grad():
        coarse = render.render_two_stream(full["xy"], *args, face, eyes, n_samples=64, return_weights=True)
        zv = render.sample_zvals(full["xy"], full["R"], full["T"], full["Kinv"], n_samples=64)
        edges = render.importance_resample(coarse["w_face"], zv, n_fine=128)
        assert edges.shape == (1, 512 * 512, 193) and torch.isfinite(edges).all()
        assert bool((edges[..., 1:] >= edges[..., :-1]).all())
        assert bool((edges[..., 0] == zv[:, 0, :, 0]).all()) and bool((edges[..., -1] <= zv[:, 0, :, -1] + 1e-6).all())
        out = render.render_two_stream(full["xy"], *args, fine, None, n_samples=192, z_edges=edges, return_weights=True)
        assert torch.isfinite(out["feat_face"]).all()
        assert _maxabs(1.0 - out["w_face"].sum(-1), out["bg_alpha_face"]) <= 2e-5
        sub = g["ray_subset"].to(dev)
        # end-to-end chain: the HIP coarse weights (<= 1e-4 from the reference's) move the inverse-cdf positions by
        # d(cdf) / pdf x bin width; FineSample parity from the reference's OWN weights is 2e-5 (fixture test above)
        assert _maxabs(edges[:, sub, :-1], g["out_zvals"][:, 0]) <= 5e-4
        assert _maxabs(out["feat_face"][:, :, sub], g["out_feat_fine"]) <= 5 * TOL
        assert _maxabs(out["bg_alpha_face"][:, :, sub], g["out_bg_alpha_fine"]) <= 5 * TOL


@pytest.mark.parametrize("binding", ["torch_ext", "ctypes"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_packed_weight_cache_skips_the_relayout_and_never_serves_stale_weights(precision, binding, monkeypatch):
    """GnrProblem.weights_packed through PackedWeightCache: a second inference call with unchanged parameters hits (the
    re-layout kernels are skipped) and is bit-identical; an in-place update (what an optimizer step or load_state_dict
    does) or a replaced tensor misses and gives the uncached result; a training call never touches the cache."""
    monkeypatch.setenv("GNR_BINDING", binding)
    dev = _dev()
    p = _to(synth.synth_problem(64, batch=1, seed=11, ray_subset=torch.arange(512) * 7 % 4096), dev)
    face = _to(synth.hash_mlp_params("face", seed=2, density_scale=5.0), dev)
    eyes = _to(synth.hash_mlp_params("eyes", seed=2, density_scale=5.0), dev)
    cache = render.PackedWeightCache()

    def run(c, f=None):
        with torch.no_grad():
            return render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                            f or face, eyes, n_samples=64, precision=precision, weight_cache=c)

    ref = run(None)
    a, b = run(cache), run(cache)
    assert (cache.misses, cache.hits) == (1, 1)
    for k in ref:
        assert torch.equal(ref[k], a[k]) and torch.equal(ref[k], b[k]), k
    face["FeaExt_module_3.weight"].mul_(1.25)               # in place: same tensor, new version
    c, ref2 = run(cache), run(None)
    assert (cache.misses, cache.hits) == (2, 1)
    assert not torch.equal(ref["feat_face"], ref2["feat_face"])
    for k in ref2:
        assert torch.equal(ref2[k], c[k]), k
    face2 = dict(face)
    face2["RGB_layer_1.weight"] = face["RGB_layer_1.weight"] * 0.5      # another tensor object
    d, ref3 = run(cache, face2), render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"],
                                                          p["appea_code"], face2, eyes, n_samples=64, precision=precision)
    assert (cache.misses, cache.hits) == (3, 1)
    for k in ref3:
        assert torch.equal(ref3[k], d[k]), k
    assert torch.equal(run(cache, face2)["feat_face"], ref3["feat_face"]) and cache.hits == 2
    # a call that needs gradients ignores the cache
    w = face2["density_module.weight"].clone().requires_grad_(True)
    out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                   dict(face2, **{"density_module.weight": w}), eyes, n_samples=64, precision=precision,
                                   weight_cache=cache)
    out["feat_face"].sum().backward()
    assert (cache.misses, cache.hits) == (3, 2) and w.grad is not None


@pytest.mark.parametrize("precision", PRECISIONS)
def test_narrow_network_vs_reference_fixture_g1_tiny(precision):
    """mlp_hidden_nchannels < 384 runs zero-padded in the 384-wide kernels.  g1_tiny: side 16, 8 samples, hidden 32 (RGB
    branch 16), B=2, train jitter -- outputs and ALL 53 gradients committed from the reference's own modules / autograd."""
    from collections import OrderedDict
    dev = _dev()
    g = load_golden("g1_tiny")
    w = {tag: OrderedDict((k[len("w_%s." % tag):], g[k]) for k in g if k.startswith("w_%s." % tag)) for tag in ("face", "eyes")}
    out, leaves, fp, ep = _grads_hip(golden_problem(g), w["face"], w["eyes"], int(g["n_samples"]), g["t_rand"], dev, precision,
                                     hidden=int(g["hidden"]))
    # 8 samples over the whole depth range (delta = 0.75 x ray length) and a x20 density head: the reference's own fp32
    # output is 4.5e-4 (bg_alpha_face) / 2.1e-4 (feat_face) away from its fp64 run on this fixture, so 5e-4 here
    tol = 5e-4
    for tag in ("face", "eyes"):
        assert _maxabs(out["feat_" + tag], g["out_feat_" + tag]) <= tol
        assert _maxabs(out["bg_alpha_" + tag], g["out_bg_alpha_" + tag]) <= tol
    for k, v in leaves.items():
        _check_grad("d" + k, v.grad, g["grad_" + k])
    for tag, params in (("face", fp), ("eyes", ep)):
        for name, v in params.items():
            assert v.grad.shape == g["gradw_%s.%s" % (tag, name)].shape
            _check_grad("%s.%s" % (tag, name), v.grad, g["gradw_%s.%s" % (tag, name)])


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("hidden", [96, 250])
def test_narrow_network_vs_oracle_live(hidden, precision):
    """Other widths (a multiple of 32 and one that is not), forward + all gradients against the oracle; and widths the
    kernels cannot hold are refused."""
    dev = _dev()
    p = synth.synth_problem(64, batch=2, camera="7", seed=21, ray_subset=torch.arange(19) * 211 % 4096)
    face = synth.hash_mlp_params("face", seed=9, hidden=hidden, density_scale=8.0)
    eyes = synth.hash_mlp_params("eyes", seed=9, hidden=hidden, density_scale=8.0)
    t_rand = synth.synth_jitter(2, 19, 40, seed=3)
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fo = {k: v.clone().requires_grad_(True) for k, v in face.items()}
    eo = {k: v.clone().requires_grad_(True) for k, v in eyes.items()}
    ref = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                              leaves["appea_code"], fo, eo, 40, t_rand=t_rand)
    O.synthetic_loss(ref).backward()
    out, hl, fp, ep = _grads_hip(p, face, eyes, 40, t_rand, dev, precision, hidden=hidden)
    for tag in ("face", "eyes"):
        assert _maxabs(out["feat_" + tag], ref["feat_" + tag]) <= TOL
        assert _maxabs(out["bg_alpha_" + tag], ref["bg_alpha_" + tag]) <= TOL
    for k in leaves:
        _check_grad("d" + k, hl[k].grad, leaves[k].grad)
    for got, want in ((fp, fo), (ep, eo)):
        for name in want:
            _check_grad(name, got[name].grad, want[name].grad)
    with pytest.raises(Exception, match="hidden"):
        big = synth.hash_mlp_params("face", seed=9, hidden=386, density_scale=8.0)
        _grads_hip(p, big, big, 40, t_rand, dev, precision, hidden=386)


@pytest.mark.parametrize("feat_nc", [6, 192, 195, 288])
def test_other_feature_widths_vs_oracle_live(feat_nc):
    """feat_nc other than the reference's 258 (opt.featmap_nc: any value 4..288 here; 3 would switch the reference to its
    sigmoid RGB head, mlp_nerf.py:116).  Round 4 computes RGB_layer_2's weight gradient as a full 192-row tile plus a
    remainder whenever feat_nc > 192: 192 (no remainder, the one-product path), 195 (3 remainder rows), 288 (the whole padded
    width) and a tiny head, forward + all gradients against the oracle."""
    dev = _dev()
    p = synth.synth_problem(64, batch=2, camera="5", seed=17, ray_subset=torch.arange(23) * 173 % 4096)
    face = synth.hash_mlp_params("face", seed=6, feat_nc=feat_nc, density_scale=8.0)
    eyes = synth.hash_mlp_params("eyes", seed=6, feat_nc=feat_nc, density_scale=8.0)
    t_rand = synth.synth_jitter(2, 23, 64, seed=4)
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fo = {k: v.clone().requires_grad_(True) for k, v in face.items()}
    eo = {k: v.clone().requires_grad_(True) for k, v in eyes.items()}
    ref = O.render_two_stream(p["xy"], leaves["R"], leaves["T"], p["Kinv"], leaves["shape_code"], leaves["gaze"],
                              leaves["appea_code"], fo, eo, 64, t_rand=t_rand)
    O.synthetic_loss(ref).backward()
    pd = _to(p, dev)
    hl = {k: pd[k].clone().requires_grad_(True) for k in leaves}
    fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
    ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
    out = render.render_two_stream(pd["xy"], hl["R"], hl["T"], pd["Kinv"], hl["shape_code"], hl["gaze"], hl["appea_code"], fp, ep,
                                   n_samples=64, t_rand=t_rand.to(dev), feat_nc=feat_nc)
    sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes")).backward()
    for tag in ("face", "eyes"):
        assert tuple(out["feat_" + tag].shape) == (2, feat_nc, 23)
        assert _maxabs(out["feat_" + tag], ref["feat_" + tag]) <= TOL
        assert _maxabs(out["bg_alpha_" + tag], ref["bg_alpha_" + tag]) <= TOL
    for k in leaves:
        _check_grad("d" + k, hl[k].grad, leaves[k].grad)
    for got, want in ((fp, fo), (ep, eo)):
        for name in want:
            _check_grad(name, got[name].grad, want[name].grad)


def _vd_weights(g):
    ds, seed = float(g["density_scale"]), int(g["weight_seed"])
    vd_ch = int(g["vd_dims"]) + synth.APPEA_DIMS
    return (synth.hash_mlp_params("face", seed=seed, vd_ch=vd_ch, density_scale=ds),
            synth.hash_mlp_params("eyes", seed=seed, vd_ch=vd_ch, density_scale=ds))


@pytest.mark.parametrize("fold", ["device", "torch"])
@pytest.mark.parametrize("binding", ["torch_ext", "ctypes"])
@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_view_direction_option_vs_reference_fixture(precision, tiled, binding, fold, monkeypatch):
    """include_vd=True (models/gaze_nerf.py:70-80, 140-143, 240-243): g11_vd holds outputs and A8-loss gradients of the
    reference's own modules with the 27-channel direction embedding in front of the appearance code.  The HIP path skips
    those weight columns and takes their per-ray fold as ``ray_bias``; the fold (plain torch) carries the gradient into the
    columns and the rotation.  tiled: the in-op ray tiling slices the per-ray bias and its gradient.
    fold="device" (round 3): no ray_bias is passed -- gnr_fwd / gnr_bwd compute the embedding, the fold, the gradient of
    the 27 weight columns and the direction's share of dR themselves (gnr_vd.hip): the C ABI alone runs include_vd."""
    from gazenerf_amd.module import VD_DIMS, view_direction_embedding
    monkeypatch.setenv("GNR_BINDING", binding)
    dev = _dev()
    g = load_golden("g11_vd")
    assert int(g["vd_dims"]) == VD_DIMS
    face, eyes = _vd_weights(g)
    pd = _to(golden_problem(g), dev)
    leaves = {k: pd[k].clone().requires_grad_(True) for k in ("R", "T", "shape_code", "gaze", "appea_code")}
    fp = {k: v.to(dev).clone().requires_grad_(True) for k, v in face.items()}
    ep = {k: v.to(dev).clone().requires_grad_(True) for k, v in eyes.items()}
    vd = view_direction_embedding(pd["xy"], leaves["R"], pd["Kinv"])
    tfold = lambda w: (torch.einsum("ok,bkr->bro", w["RGB_layer_1.weight"].reshape(192, -1)[:, 384:384 + VD_DIMS], vd).contiguous()
                       if fold == "torch" else None)
    out = render.render_two_stream(pd["xy"], leaves["R"], leaves["T"], pd["Kinv"], leaves["shape_code"], leaves["gaze"],
                                   leaves["appea_code"], fp, ep, n_samples=int(g["n_samples"]), t_rand=g["t_rand"].to(dev),
                                   precision=precision, vd_dims=VD_DIMS, ray_bias_face=tfold(fp), ray_bias_eyes=tfold(ep),
                                   ray_tile=8 if tiled else None)
    for tag in ("face", "eyes"):
        assert _maxabs(out["feat_" + tag], g["out_feat_" + tag]) <= TOL
        assert _maxabs(out["bg_alpha_" + tag], g["out_bg_alpha_" + tag]) <= TOL
    (sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))).backward()
    for k, v in leaves.items():
        _check_grad("d" + k, v.grad, g["grad_" + k])
    for tag, params in (("face", fp), ("eyes", ep)):
        for name, v in params.items():
            ref = g["gradw_%s.%s" % (tag, name)]
            got = v.grad
            if name == "RGB_layer_1.weight":
                _check_grad(tag + " view-direction columns", got.reshape(192, -1)[:, 384:384 + VD_DIMS], g["gradw_vdcols_" + tag])
            if got.numel() > 4096:
                got = got.reshape(got.shape[0], -1)[::16]
            _check_grad("%s.%s" % (tag, name), got.reshape(ref.shape), ref)


def test_module_with_view_direction_matches_the_oracle():
    """HotPathRenderer(include_vd=True): parameter shapes of the reference (RGB_layer_1 [192, 384+27+127, 1, 1]) and the
    oracle's include_vd forward on the module's own weights."""
    from gazenerf_amd import HotPathRenderer
    dev = _dev()
    net = HotPathRenderer(include_vd=True).to(dev)
    assert tuple(net.fg_CD_predictor_face.RGB_layer_1.weight.shape) == (192, 384 + 27 + 127, 1, 1)
    p = synth.synth_problem(64, batch=2, camera="2", seed=4, ray_subset=torch.arange(40) * 97 % 4096)
    pd = _to(p, dev)
    with torch.no_grad():
        got = net(pd["xy"], pd["R"], pd["T"], pd["Kinv"], pd["shape_code"], pd["appea_code"], pd["gaze"])
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.named_parameters()}
        ref = O.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"],
                                  cpu(net.fg_CD_predictor_face), cpu(net.fg_CD_predictor_eyes), 64, include_vd=True)
    for tag in ("face", "eyes"):
        assert _maxabs(got["feat_" + tag], ref["feat_" + tag]) <= TOL
        assert _maxabs(got["bg_alpha_" + tag], ref["bg_alpha_" + tag]) <= TOL
