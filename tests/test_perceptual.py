"""SURVEY.md 8(f) N4, the VGG-perceptual and gaze-angular terms: gazenerf_amd.perceptual against values captured from the
reference's own ``VGGPerceptualLoss``, ``GazePerceptualLoss.forward`` + ``gaze_network`` and ``GazeNeRFLoss.calc_total_loss`` with
every term switched on, run on this repository's restatement of torchvision's ``vgg16().features`` layout with hashed weights
(oracle/gen_golden_vgg.py).  The loss arithmetic is pinned, the
architecture is restated from torchvision's published configuration (not importable offline), the ImageNet weights are the
caller's to supply."""
import pytest
import torch

from conftest import load_golden
from gazenerf_amd import gan as G
from gazenerf_amd import losses as L
from gazenerf_amd import perceptual as P

VGG_SEED, SEED_X = 5, 53


def _vgg(dev="cpu"):
    f = P.vgg16_features()
    f.load_state_dict(P.hash_vgg16_state(VGG_SEED))
    return P.VGGPerceptualLoss(resize=True, features=f).to(dev)


def _gaze(dev="cpu"):
    f = P.vgg16_features_full()
    f.load_state_dict(P.hash_vgg16_state(VGG_SEED, full=True))
    net = P.GazeNetwork(f)
    head = P.hash_gaze_head_state(seed=2)
    head["FC3.weight"] = head["FC3.weight"] * 0.5            # as the generator: estimates a few degrees apart, tanh unsaturated
    net.load_state_dict(head, strict=False)
    return P.GazeAngularLoss(model=net).to(dev)


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def test_layout_is_torchvisions_vgg16_features_up_to_index_22():
    f = P.vgg16_features()
    conv = {i: (m.in_channels, m.out_channels) for i, m in enumerate(f) if isinstance(m, torch.nn.Conv2d)}
    assert conv == {0: (3, 64), 2: (64, 64), 5: (64, 128), 7: (128, 128), 10: (128, 256), 12: (256, 256), 14: (256, 256),
                    17: (256, 512), 19: (512, 512), 21: (512, 512)}
    assert [i for i, m in enumerate(f) if isinstance(m, torch.nn.MaxPool2d)] == [4, 9, 16] and len(f) == 23
    v = P.VGGPerceptualLoss()
    assert not v.pretrained and not any(q.requires_grad for q in v.parameters())
    # the reference's module: blocks = features[:4], [4:9], [9:16], [16:23]; Sequential slices keep torchvision's indices
    keys = [k for k in v.state_dict() if k.startswith("blocks")]
    assert keys[:4] == ["blocks.0.0.weight", "blocks.0.0.bias", "blocks.0.2.weight", "blocks.0.2.bias"]
    assert "blocks.1.5.weight" in keys and "blocks.2.14.bias" in keys and "blocks.3.21.weight" in keys and len(keys) == 20
    assert v(torch.rand(1, 3, 40, 40), torch.rand(1, 3, 40, 40)).ndim == 0


def test_torchvision_checkpoint_keys_load_and_malformed_ones_are_refused():
    state = {"features." + k: v for k, v in P.hash_vgg16_state(1).items()}
    state["features.24.weight"] = torch.zeros(512, 512, 3, 3)           # deeper layers and the classifier are ignored
    state["classifier.0.weight"] = torch.zeros(8, 8)
    v = P.VGGPerceptualLoss(pretrained_state=state)
    assert v.pretrained and torch.equal(v.blocks[1]._modules["5"].weight, state["features.5.weight"])     # slices keep torchvision's indices as names
    bad = dict(state); del bad["features.19.bias"]
    with pytest.raises(KeyError):
        P.VGGPerceptualLoss(pretrained_state=bad)
    bad = dict(state); bad["features.0.weight"] = torch.zeros(64, 3, 5, 5)
    with pytest.raises(ValueError):
        P.VGGPerceptualLoss(pretrained_state=bad)


def _check_module(dev, tol, tol_grad):
    g = load_golden("g14_vgg")
    case = G.synth_gan_case(seed=SEED_X)
    v = _vgg(dev)
    x0, y = case["fake_img"][:, :, ::2, ::2].to(dev), case["real_img"][:, :, ::2, ::2].to(dev)
    for tag, kw in {"all": {}, "style": {"feature_layers": [1, 3], "style_layers": [0, 2]}}.items():
        x = x0.clone().requires_grad_(True)
        out = v(x, y, **kw)
        assert abs(float(out) - g["module_" + tag]) <= tol * abs(g["module_" + tag]), tag
        out.backward()
        e = _rel(x.grad[:, :, ::4, ::4], g["module_%s_grad" % tag])
        assert e <= tol_grad, (tag, e)
    assert abs(float(v(x0[:, :1], y[:, :1])) - g["module_gray"]) <= tol * abs(g["module_gray"])


def _check_gaze(dev, tol, tol_grad):
    g = load_golden("g14_vgg")
    case = {k: t.to(dev) for k, t in G.synth_gan_case(seed=SEED_X).items()}
    gz = _gaze(dev)
    with torch.no_grad():
        gaze, head = gz.model((case["real_img"] - gz.mean) / gz.std)
    assert float((gaze.cpu() - g["gaze_pitchyaw"]).abs().max()) <= tol and float((head.cpu() - g["head_pitchyaw"]).abs().max()) <= tol
    x = case["fake_img"].clone().requires_grad_(True)
    out = gz(x, case["real_img"])
    assert abs(float(out) - g["angular_deg"]) <= 50 * tol * g["angular_deg"]      # degrees: acos amplifies the estimates' rounding
    out.backward()
    e = _rel(x.grad[:, :, ::8, ::8], g["angular_grad"])
    assert e <= tol_grad, e


def _check_total(dev, tol, tol_grad):
    g = load_golden("g14_vgg")
    case = {k: t.to(dev) for k, t in G.synth_gan_case(seed=SEED_X).items()}
    v = _vgg(dev)
    gz = _gaze(dev)
    d = G.PatchGAN(3, 8)
    d.load_state_dict(G.hash_patchgan_state(seed=3, ndf=8))
    d = d.to(dev).train()
    for q in d.parameters():
        q.requires_grad = False
    pred = {k: case[k].clone().requires_grad_(True) for k in ("merge_img_face", "merge_img_eyes", "bg_img")}
    pred["merge_img"] = case["fake_img"].clone().requires_grad_(True)
    codes = {"bg": None, "iden": case["code_iden"], "expr": case["code_expr"], "appea": case["code_appea"]}
    out = L.total_loss(pred, case["gt"], case["face"], case["full_eye"], case["leye"], case["reye"], codes, None, use_l1=True,
                       epoch=1, discriminator=d, batch_num=3, vgg=v, vgg_importance=0.7, gaze=gz, eye_loss_importance=31.0)
    want = [k[6:] for k in g if k.startswith("total_") and not k.startswith("total_grad_")]
    # the reference sums its dictionary in insertion order: the terms must come in ITS order
    assert [k for k in out if k in want] == ["bg_loss", "eyes_loss", "face_loss", "nonhead_loss", "head_loss", "vgg_face_loss",
                                              "vgg_eyes_loss", "vgg", "angular", "gen_patch_gan_loss", "iden_code", "expr_code",
                                              "appea_code", "bg_code", "total_loss"]
    for k in want:
        assert abs(float(out[k]) - g["total_" + k]) <= tol * max(1.0, abs(g["total_" + k])), k
    out["total_loss"].backward()
    for k, t in pred.items():
        e = _rel(t.grad[:, :, ::8, ::8], g["total_grad_" + k])
        assert e <= tol_grad, (k, e)


def test_module_vs_reference_fixture():
    torch.set_num_threads(1)
    _check_module("cpu", 1e-5, 1e-4)


def test_gaze_network_and_angular_loss_vs_reference_fixture():
    torch.set_num_threads(1)
    _check_gaze("cpu", 1e-6, 1e-4)


def test_gaze_network_has_the_reference_checkpoints_names():
    keys = list(P.GazeNetwork().state_dict().keys())
    assert keys[:2] == ["vgg16.0.weight", "vgg16.0.bias"] and keys[-6:] == ["FC1.weight", "FC1.bias", "FC2.weight", "FC2.bias", "FC3.weight", "FC3.bias"]
    assert len(keys) == 26 + 6 and len(P.vgg16_features_full()) == 31
    a = torch.tensor([[0.1, -0.2], [0.0, 0.0]])
    v = P.pitchyaw_to_vector(a)
    assert float((v.norm(dim=1) - 1).abs().max()) <= 1e-6 and torch.allclose(v[1], torch.tensor([0.0, 0.0, 1.0]))
    assert float(P.angular_distance_deg(v, v).abs().max()) <= 0.1            # acos at 1: sqrt(eps)-sized, in degrees
    assert abs(float(P.angular_distance_deg(v[:1], -v[:1])) - 180.0) <= 0.1


def test_total_loss_with_every_term_vs_reference_fixture():
    torch.set_num_threads(1)
    _check_total("cpu", 1e-5, 1e-4)
    # the angular term only exists from epoch 0 on (gazenerf_loss.py:383), like head_loss
    case = G.synth_gan_case(seed=SEED_X, side=32)
    pred = {k: case[k] for k in ("merge_img_face", "merge_img_eyes", "bg_img")}
    pred["merge_img"] = case["fake_img"]
    codes = {"bg": None, "iden": case["code_iden"], "expr": case["code_expr"], "appea": case["code_appea"]}
    out = L.total_loss(pred, case["gt"], case["face"], case["full_eye"], case["leye"], case["reye"], codes, None, epoch=-1,
                       gaze=P.GazeAngularLoss())
    assert "angular" not in out and "head_loss" not in out


@pytest.mark.gpu
def test_perceptual_terms_on_gpu_vs_reference_fixture():
    """Through MIOpen on the MI355X: loss values 1e-4 relative; the image gradients pass ten ReLUs and three max-pools whose
    kinks another summation order crosses here and there (millions of elements, each worth ~1e-3 of the norm at most): 2e-2."""
    _check_module(torch.device("cuda:0"), 1e-4, 2e-2)
    _check_gaze(torch.device("cuda:0"), 1e-4, 5e-2)
    _check_total(torch.device("cuda:0"), 1e-4, 2e-2)
