"""ORACLE tooling -- test infrastructure, NOT product code.

Fixture for the VGG-perceptual and gaze-angular terms of SURVEY.md 8(f) N4: the reference's own ``VGGPerceptualLoss``
(losses/gazenerf_loss.py:40-102), ``GazePerceptualLoss.forward`` (:160-190) on its own ``gaze_network``
(gaze_estimation/xgaze_baseline_vgg.py) and ``GazeNeRFLoss.calc_total_loss`` with every term switched on (:360-401) evaluated here.
``GazePerceptualLoss.__init__`` reads a checkpoint and cv2 camera files that do not exist offline and that ``forward`` never uses:
the object is assembled without it (``__new__`` + the attributes ``forward`` reads), its ``forward`` is the reference's.  torchvision is
not installed and the ImageNet weights are not available: ``torchvision.models.vgg16`` is replaced by a factory returning
gazenerf_amd.perceptual's restatement of the ``features`` layout with hashed weights -- so what this pins is the reference's loss
ARITHMETIC on a given feature extractor (normalisation, resize, the four slices, L1 sums, target masking, term weights), not
torchvision's architecture.  Inputs are hashed too; only expected outputs are stored.  Writes tests/golden/g14_vgg.npz.

    python oracle/gen_golden_vgg.py
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


def main():
    sys.dont_write_bytecode = True
    from gazenerf_amd import gan as G
    from gazenerf_amd import losses as L
    from gazenerf_amd import perceptual as P

    for name in ("cv2", "torchvision", "gaze_estimation", "gaze_estimation.xgaze_baseline_vgg", "wandb", "imageio",
                 "skimage", "skimage.metrics", "piq", "kornia", "kornia.filters", "h5py", "lpips", "face_recognition"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    tr = types.ModuleType("torchvision.transforms")
    ident = lambda *a, **k: (lambda x: x)
    tr.Compose, tr.ToPILImage, tr.ToTensor, tr.Normalize, tr.Resize = (lambda fs: (lambda x: x)), ident, ident, ident, ident
    tv.transforms = tr
    sys.modules["torchvision.transforms"] = tr
    VGG_SEED = 5

    def fake_vgg16(pretrained=False):
        f = P.vgg16_features_full()              # the perceptual loss slices [:23] of it, the gaze estimator takes all 31 modules
        f.load_state_dict(P.hash_vgg16_state(VGG_SEED, full=True))
        return types.SimpleNamespace(features=f)
    tv.models = types.SimpleNamespace(vgg16=fake_vgg16)
    sys.modules["torchvision.models"] = tv.models
    del sys.modules["gaze_estimation"], sys.modules["gaze_estimation.xgaze_baseline_vgg"]      # the real package: its gaze_network is used
    sys.path.insert(0, REF)
    os.chdir(REF)
    from losses.gazenerf_loss import GazeNeRFLoss, GazePerceptualLoss, VGGPerceptualLoss
    from models.discriminator import PatchGAN as RefPatchGAN

    torch.set_num_threads(1)
    B, S = 2, 224
    case = G.synth_gan_case(seed=53, batch=B, side=S)
    arrays = {}

    # 1. the module alone: all four blocks, then a subset with a style (Gram) term
    ref_v = VGGPerceptualLoss(resize=True)
    feats = P.vgg16_features()
    feats.load_state_dict(P.hash_vgg16_state(VGG_SEED))
    our_v = P.VGGPerceptualLoss(resize=True, features=feats)
    assert [k for k in ref_v.state_dict()] == [k for k in our_v.state_dict()]            # blocks.<i>.<index>.weight ..., mean, std
    x = case["fake_img"][:, :, ::2, ::2].clone().requires_grad_(True)                    # 112 x 112: the resize is exercised
    y = case["real_img"][:, :, ::2, ::2]
    for tag, kw in {"all": {}, "style": {"feature_layers": [1, 3], "style_layers": [0, 2]}}.items():
        x.grad = None
        a = ref_v(x, y, **kw)
        a.backward()
        ga = x.grad.clone()
        x.grad = None
        b = our_v(x, y, **kw)
        b.backward()
        e = float((ga - x.grad).norm() / ga.norm())
        print("  module %-5s ref %.8f ours %.8f  d/dx rel-L2 diff %.2e" % (tag, float(a), float(b), e))
        assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a)) and e <= 1e-6
        arrays["module_" + tag] = np.float64(float(a))
        arrays["module_%s_grad" % tag] = ga[:, :, ::4, ::4].clone()
    # one-channel inputs are repeated to three (gazenerf_loss.py:77-79)
    a, b = ref_v(x[:, :1].detach(), y[:, :1]), our_v(x[:, :1].detach(), y[:, :1])
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))
    arrays["module_gray"] = np.float64(float(a))

    # 2. the gaze-angular loss: the reference's forward on its own gaze_network, hashed weights (head scaled so that the two
    #    images' gaze estimates differ by degrees: away from acos's singular end and from tanh's saturation)
    from gaze_estimation.xgaze_baseline_vgg import gaze_network
    head = P.hash_gaze_head_state(seed=2)
    head["FC3.weight"] = head["FC3.weight"] * 0.5
    ref_g = GazePerceptualLoss.__new__(GazePerceptualLoss)
    torch.nn.Module.__init__(ref_g)
    ref_g.model = gaze_network()
    ref_g.model.load_state_dict(head, strict=False)
    ref_g.model.eval()
    ref_g.resize, ref_g.device = True, "cpu"
    ref_g.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
    ref_g.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))
    our_net = P.GazeNetwork()
    assert list(our_net.state_dict().keys()) == list(ref_g.model.state_dict().keys())            # vgg16.<i>.*, FC1-3: the checkpoint's names
    our_net.load_state_dict(ref_g.model.state_dict(), strict=True)
    our_g = P.GazeAngularLoss(model=our_net)
    xg = case["fake_img"].clone().requires_grad_(True)           # 224 x 224: trans_eval (stubbed to the identity) == our resize
    a = ref_g(xg, case["real_img"], None, [None])
    a.backward()
    ga = xg.grad.clone()
    xg.grad = None
    b = our_g(xg, case["real_img"])
    b.backward()
    e = float((ga - xg.grad).norm() / ga.norm())
    print("  angular ref %.6f deg ours %.6f deg  d/dx rel-L2 diff %.2e" % (float(a), float(b), e))
    assert float(a) > 0.5 and abs(float(a) - float(b)) <= 1e-6 * float(a) and e <= 1e-6
    arrays["angular_deg"] = np.float64(float(a))
    arrays["angular_grad"] = ga[:, :, ::8, ::8].clone()
    with torch.no_grad():
        gz, hd = ref_g.model((case["real_img"] - ref_g.mean) / ref_g.std)
    arrays["gaze_pitchyaw"], arrays["head_pitchyaw"] = gz.clone(), hd.clone()

    # 3. inside calc_total_loss: use_vgg_loss, use_angular_loss and use_patch_gan_loss together
    NDF = 8
    ref_d = RefPatchGAN(input_nc=3, ndf=NDF)
    ref_d.load_state_dict(G.hash_patchgan_state(seed=3, ndf=NDF))
    our_d = G.PatchGAN(3, NDF)
    our_d.load_state_dict(ref_d.state_dict())
    for d in (ref_d, our_d):
        d.train()
        for q in d.parameters():
            q.requires_grad = False
    pred = {k: case[k].clone().requires_grad_(True) for k in ("merge_img_face", "merge_img_eyes", "bg_img")}
    pred["merge_img"] = case["fake_img"].clone().requires_grad_(True)
    gt, face, leye, reye, full_eye = (case[k] for k in ("gt", "face", "leye", "reye", "full_eye"))
    codes = {"bg": None, "iden": case["code_iden"], "expr": case["code_expr"], "appea": case["code_appea"]}
    loss = GazeNeRFLoss(eye_loss_importance=1.0, vgg_importance=0.7, use_vgg_loss=True, use_l1_loss=True,
                        use_patch_gan_loss=True, device="cpu")
    loss.eye_loss_importance = 31.0            # its value after one increase_eye_importance() (:256-262)
    loss.use_angular_loss, loss.gaze_loss_func = True, ref_g        # what the constructor does with use_angular_loss=True (:233-236)
    rl = loss.calc_total_loss(delta_cam_info=None, opt_code_dict=codes, pred_dict={"coarse_dict": pred}, gt_rgb=gt,
                              face_mask_tensor=face, full_eye_mask_tensor=full_eye, left_eye_mask_tensor=leye,
                              right_eye_mask_tensor=reye, cam_ind=None, ldms=[None], epoch=1, batch_num=3, discriminator=ref_d)
    rl["total_loss"].backward()
    gref = {k: v.grad.clone() for k, v in pred.items() if v.grad is not None}
    for v in pred.values():
        v.grad = None
    ol = L.total_loss(pred, gt, face, full_eye, leye, reye, codes, None, use_l1=True, epoch=1, discriminator=our_d, batch_num=3,
                      vgg=our_v, vgg_importance=0.7, gaze=our_g, eye_loss_importance=31.0)
    ol["total_loss"].backward()
    assert list(rl.keys()) == list(ol.keys()), (list(rl.keys()), list(ol.keys()))       # same terms in the same order
    for k in rl:
        print("  total %-20s ref %.8f ours %.8f" % (k, float(rl[k]), float(ol[k])))
        assert abs(float(rl[k]) - float(ol[k])) <= 2e-6 * max(1.0, abs(float(rl[k]))), k
        arrays["total_" + k] = np.float64(float(rl[k]))
    for k, gr in gref.items():
        e = float((gr - pred[k].grad).norm() / gr.norm())
        print("  d total / d %-15s rel-L2 diff %.2e" % (k, e))
        assert e <= 1e-5, k
        arrays["total_grad_" + k] = gr[:, :, ::8, ::8].clone()
    np.savez_compressed(os.path.join(GOLD, "g14_vgg.npz"), **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print("  wrote tests/golden/g14_vgg.npz (%.0f KB)" % (os.path.getsize(os.path.join(GOLD, "g14_vgg.npz")) / 1024))


if __name__ == "__main__":
    main()
