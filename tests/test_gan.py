"""SURVEY.md 8(f) N4, the PatchGAN terms: gazenerf_amd.gan against values captured from the reference's own ``PatchGAN``,
``discriminator_loss`` / ``generator_loss`` and ``GazeNeRFLoss.calc_total_loss(use_patch_gan_loss=True)``
(oracle/gen_golden_gan.py: hashed parameters and inputs, only expected outputs are stored), and the trainer's
discriminator-then-generator step (trainer/gazenerf_trainer.py:487-528)."""
import pytest
import torch

from conftest import load_golden
from gazenerf_amd import gan as G
from gazenerf_amd import losses as L

NDF, SEED_W, SEED_X = 8, 3, 53        # input seed: the widest margin around the LeakyReLU kinks (oracle/gen_golden_gan.py)


def _setup(dev="cpu"):
    d = G.PatchGAN(3, NDF)
    d.load_state_dict(G.hash_patchgan_state(seed=SEED_W, ndf=NDF), strict=True)
    d.train()
    case = {k: v.to(dev) for k, v in G.synth_gan_case(seed=SEED_X).items()}
    return d.to(dev), case


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def test_state_dict_surface_is_the_reference_discriminators():
    keys = list(G.PatchGAN(3, 64).state_dict().keys())
    # models/discriminator.py:10-22 in definition order
    want = ["conv1.weight", "conv1.bias", "conv2.weight"] + ["norm1." + k for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")] + \
           ["conv3.weight"] + ["norm2." + k for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")] + \
           ["conv4.weight"] + ["norm3." + k for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")] + \
           ["conv5.weight", "conv5.bias"]
    assert keys == want
    d = G.PatchGAN(3, 64)
    assert d.conv1.weight.shape == (64, 3, 6, 6) and d.conv4.weight.shape == (512, 256, 6, 6) and d.conv5.weight.shape == (1, 512, 6, 6)
    assert d(torch.zeros(1, 3, 224, 224)).shape == (1, 1, 20, 20)


def _check_discriminator_side(dev, tol_logit, tol_rel, tol_rel_upstream=None):
    """tol_rel_upstream: bound for the gradients UPSTREAM of a LeakyReLU (everything but conv5's) -- one pre-activation that
    another summation order puts on the other side of zero moves them by ~0.7 % (gazenerf_amd.gan.PatchGAN docstring)."""
    tol_rel_upstream = tol_rel if tol_rel_upstream is None else tol_rel_upstream
    g = load_golden("g13_patchgan")
    d, case = _setup(dev)
    real, fake = d(case["real_img"]), d(case["fake_img"])
    assert float((real.detach().cpu() - g["logits_real"]).abs().max()) <= tol_logit
    assert float((fake.detach().cpu() - g["logits_fake"]).abs().max()) <= tol_logit
    dl, gl = G.discriminator_loss(real, fake), G.generator_loss(fake)
    assert abs(float(dl) - g["disc_loss"]) <= tol_logit and abs(float(gl) - g["gen_loss"]) <= tol_logit
    dl.backward()
    grads = dict(d.named_parameters())
    for k in g:
        if k.startswith("dgrad_"):
            e = _rel(grads[k[6:]].grad, g[k])
            assert e <= (tol_rel if k.startswith("dgrad_conv5") else tol_rel_upstream), (k, e)
    assert _rel(d.norm1.running_mean, g["norm1_running_mean"]) <= tol_rel
    assert _rel(d.norm3.running_var, g["norm3_running_var"]) <= tol_rel


def _check_generator_side(dev, tol, tol_rel):       # tol_rel: d total / d image passes all four LeakyReLUs
    g = load_golden("g13_patchgan")
    d, case = _setup(dev)
    d(case["real_img"]); d(case["fake_img"])             # the two forwards of the discriminator step move the running statistics
    for q in d.parameters():
        q.requires_grad = False
    pred = {k: case[k] for k in ("merge_img_face", "merge_img_eyes", "bg_img")}
    codes = {"bg": None, "iden": case["code_iden"], "expr": case["code_expr"], "appea": case["code_appea"]}
    for tag in ("ramp", "full"):
        epoch, batch_num = (int(v) for v in g[tag + "_epoch_batch"])
        img = case["fake_img"].clone().requires_grad_(True)
        pred["merge_img"] = img
        out = L.total_loss(pred, case["gt"], case["face"], case["full_eye"], case["leye"], case["reye"], codes, None,
                           epoch=epoch, discriminator=d, batch_num=batch_num)
        assert abs(float(out["gen_patch_gan_loss"]) - g[tag + "_gen_patch_gan_loss"]) <= tol
        assert abs(float(out["total_loss"]) - g[tag + "_total_loss"]) <= 10 * tol
        out["total_loss"].backward()
        e = _rel(img.grad[:, :, ::8, ::8], g[tag + "_grad_merge_img"])
        assert e <= tol_rel, (tag, e)
        assert all(q.grad is None for q in d.parameters())          # frozen: the generator step leaves no gradient in D


def test_discriminator_vs_reference_fixture():
    torch.set_num_threads(1)
    _check_discriminator_side("cpu", 1e-5, 1e-4)


def test_generator_term_and_total_loss_vs_reference_fixture():
    torch.set_num_threads(1)
    _check_generator_side("cpu", 1e-6, 1e-4)


def test_warm_up_ramp():
    # gazenerf_loss.py:398
    assert G.warm_up_coeff(0, 0) == 0.0 and G.warm_up_coeff(-1, 0) == 0.0
    assert abs(G.warm_up_coeff(0, 5000) - 0.025) < 1e-12 and G.warm_up_coeff(0, 20000) == 0.1
    assert G.warm_up_coeff(0, 150000) == 0.1 and G.warm_up_coeff(3, 7) == 0.1


def test_resize_is_the_identity_at_224_and_bilinear_otherwise():
    x = torch.rand(1, 3, 224, 224)
    assert torch.equal(G.resize_224(x), x)
    y = G.resize_224(torch.rand(1, 3, 512, 512))
    assert y.shape == (1, 3, 224, 224)
    ramp = torch.linspace(0, 1, 448).view(1, 1, 1, 448).expand(1, 3, 448, 448)
    # a linear ramp halved: output pixel i = mean of input pixels 2i, 2i+1 (align_corners=False, no antialias needed at 2x)
    want = (ramp[..., 0::2] + ramp[..., 1::2])[:, :, ::2] / 2
    assert float((G.resize_224(ramp) - want).abs().max()) <= 1e-6


def test_discriminator_step_trains_and_leaves_the_parameters_frozen():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    step = G.DiscriminatorStep("cpu", lr=2e-3, ndf=4)
    case = G.synth_gan_case(seed=5, batch=2, side=64)
    gt = case["gt"] * 0.2 + 0.7                              # "real" patches differ from the generated ones in their statistics
    before = [q.detach().clone() for q in step.discriminator.parameters()]
    hist = [float(step.step(gt, case["face"], case["fake_img"])["disc_loss"]) for _ in range(12)]
    assert hist[-1] < hist[0], hist
    assert all(not q.requires_grad for q in step.discriminator.parameters())
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(before, step.discriminator.parameters()))
    img = case["fake_img"].clone().requires_grad_(True)
    step.generator_term(img, epoch=1, batch_num=0).backward()
    assert float(img.grad.abs().sum()) > 0


@pytest.mark.gpu
def test_patchgan_on_gpu_vs_reference_fixture():
    """The same fixture through MIOpen's convolutions on the MI355X (fp32): logits and losses 1e-4, conv5's gradients 1e-3
    rel-L2; the gradients upstream of the LeakyReLUs within 3e-2 -- room for a pre-activation that MIOpen's Winograd backward
    rounds to the other side of zero (the fixture's margin is 6.9e-6; measured on the round's boxes: no flip, <= 3e-6)."""
    _check_discriminator_side(torch.device("cuda:0"), 1e-4, 1e-3, 3e-2)
    _check_generator_side(torch.device("cuda:0"), 1e-5, 3e-2)


@pytest.mark.gpu
def test_fitter_step_with_every_loss_term():
    """trainer/gazenerf_trainer.py:487-528 around the whole network with use_vgg_loss, use_angular_loss and use_patch_gan_loss: discriminator update on (ground truth, detached
    prediction), then the generator step with the PatchGAN term in its total loss; D stays frozen during the latter."""
    from gazenerf_amd import GazeNeRFNetAMD, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = GazeNeRFNetAMD(featmap_size=32, pred_img_size=256, num_sample_coarse=32).to(dev)
    B, S = 2, 256
    p = {k: v.to(dev) for k, v in synth.synth_problem(32, batch=B, camera="5", seed=2).items()}
    base = {"iden": p["shape_code"][:, :100], "expr": p["shape_code"][:, 100:], "text": p["appea_code"][:, :100],
            "illu": p["appea_code"][:, 100:], "gaze": p["gaze"], "c2w_Rmat": p["R"], "c2w_Tvec": p["T"], "inv_inmat": p["Kinv"]}
    yy, xx = torch.meshgrid(torch.arange(S, device=dev), torch.arange(S, device=dev), indexing="ij")
    disk = lambda cy, cx, r: (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float().view(1, 1, S, S).expand(B, 1, S, S)
    face, leye, reye = disk(128, 128, 96), disk(104, 88, 18), disk(104, 168, 18)
    full_eye = torch.clamp(leye + reye, 0, 1)
    gt = torch.rand(B, 3, S, S, device=dev) * face + (1.0 - face)
    fit = L.Fitter(net, n_rows=2, lr=1e-3)
    gan = G.DiscriminatorStep(dev, lr=1e-3, ndf=16)
    from gazenerf_amd import perceptual as P
    vgg, gaze_loss = P.VGGPerceptualLoss().to(dev), P.GazeAngularLoss().to(dev)       # random features: the code path, not the reference's loss
    d0 = [q.detach().clone() for q in gan.discriminator.parameters()]
    out = None
    for i in range(3):
        t_rand = synth.synth_jitter(B, 32 * 32, 32, seed=i).to(dev)
        out = fit.step(slice(0, B), p["xy"], base, gt, face, full_eye, leye, reye, t_rand=t_rand, epoch=1, gan=gan, batch_num=i,
                       vgg=vgg, vgg_importance=0.5, gaze_loss=gaze_loss, eye_loss_importance=31.0)
    assert {"gen_patch_gan_loss", "disc_loss", "total_loss", "vgg_face_loss", "vgg_eyes_loss", "vgg", "angular"} <= set(out)
    assert out["gen_patch_gan_loss"] > 0 and all(v == v for v in out.values())
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(d0, gan.discriminator.parameters()))
    assert all(not q.requires_grad for q in gan.discriminator.parameters())      # frozen again after the generator step
