"""ORACLE tooling -- test infrastructure, NOT product code.

Fixture for the VGG-perceptual terms of SURVEY.md 8(f) N4: the reference's own ``VGGPerceptualLoss`` (losses/gazenerf_loss.py:40-102)
and ``GazeNeRFLoss.calc_total_loss`` with ``use_vgg_loss=True`` (+ the PatchGAN term, :360-401) evaluated here.  torchvision is
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
        f = P.vgg16_features()
        f.load_state_dict(P.hash_vgg16_state(VGG_SEED))
        return types.SimpleNamespace(features=f)
    tv.models = types.SimpleNamespace(vgg16=fake_vgg16)
    sys.modules["gaze_estimation.xgaze_baseline_vgg"].gaze_network = object
    sys.path.insert(0, REF)
    os.chdir(REF)
    from losses.gazenerf_loss import GazeNeRFLoss, VGGPerceptualLoss
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

    # 2. inside calc_total_loss: use_vgg_loss=True together with the PatchGAN term
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
    rl = loss.calc_total_loss(delta_cam_info=None, opt_code_dict=codes, pred_dict={"coarse_dict": pred}, gt_rgb=gt,
                              face_mask_tensor=face, full_eye_mask_tensor=full_eye, left_eye_mask_tensor=leye,
                              right_eye_mask_tensor=reye, cam_ind=None, ldms=None, epoch=1, batch_num=3, discriminator=ref_d)
    rl["total_loss"].backward()
    gref = {k: v.grad.clone() for k, v in pred.items() if v.grad is not None}
    for v in pred.values():
        v.grad = None
    ol = L.total_loss(pred, gt, face, full_eye, leye, reye, codes, None, use_l1=True, epoch=1, discriminator=our_d, batch_num=3,
                      vgg=our_v, vgg_importance=0.7)
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
