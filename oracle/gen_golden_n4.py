"""ORACLE tooling -- test infrastructure, NOT product code.

Fixture for SURVEY.md 8(f) N4 (the terms that need no pretrained network): the reference's own
``GazeNeRFLoss.calc_total_loss`` (losses/gazenerf_loss.py:405-470, constructed with use_vgg_loss=False) and
``BaseTrainer.eulurangle2Rmat`` (trainer/base.py:92-124) evaluated here on seeded inputs.  torchvision / cv2 /
the gaze-estimator package are import-time dependencies of that file only; they are replaced by empty modules.
Writes tests/golden/g10_losses.npz.

    python oracle/gen_golden_n4.py
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
    for name in ("cv2", "torchvision", "gaze_estimation", "gaze_estimation.xgaze_baseline_vgg", "wandb", "imageio",
                 "skimage", "skimage.metrics", "piq", "kornia", "kornia.filters", "h5py", "lpips", "face_recognition"):
        sys.modules.setdefault(name, types.ModuleType(name))
    tv = sys.modules["torchvision"]
    tr = types.ModuleType("torchvision.transforms")
    ident = lambda *a, **k: (lambda x: x)
    tr.Compose, tr.ToPILImage, tr.ToTensor, tr.Normalize, tr.Resize = (lambda fs: (lambda x: x)), ident, ident, ident, ident
    tv.transforms = tr
    sys.modules["torchvision.transforms"] = tr
    sys.modules["gaze_estimation.xgaze_baseline_vgg"].gaze_network = object
    sys.path.insert(0, REF)
    os.chdir(REF)
    from losses.gazenerf_loss import GazeNeRFLoss
    from trainer.base import BaseTrainer

    from gazenerf_amd import losses as L

    g = torch.Generator().manual_seed(7)
    B, S = 2, 64
    rnd = lambda *s: torch.rand(*s, generator=g)
    pred = {k: rnd(B, 3, S, S) for k in ("merge_img_face", "merge_img_eyes", "merge_img", "bg_img")}
    gt = rnd(B, 3, S, S)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    disk = lambda cy, cx, r: (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float().view(1, 1, S, S).expand(B, 1, S, S).clone()
    face, leye, reye = disk(32, 32, 24), disk(26, 22, 5), disk(26, 42, 5)
    full_eye = torch.clamp(leye + reye, 0, 1)
    codes = {"bg": None, "iden": 0.1 * torch.randn(B, 100, generator=g), "expr": 0.1 * torch.randn(B, 79, generator=g),
             "appea": 0.1 * torch.randn(B, 127, generator=g)}
    delta = {"delta_eulur": 0.05 * torch.randn(B, 3, generator=g), "delta_tvec": 0.05 * torch.randn(B, 3, 1, generator=g)}
    arrays = {"gt": gt, "face": face, "leye": leye, "reye": reye, "full_eye": full_eye,
              **{"pred_" + k: v for k, v in pred.items()}, **{"code_" + k: v for k, v in codes.items() if v is not None},
              **delta}
    for use_l1 in (False, True):
        ref = GazeNeRFLoss(eye_loss_importance=1.0, vgg_importance=1.0, use_vgg_loss=False, use_l1_loss=use_l1)
        rl = ref.calc_total_loss(delta_cam_info=delta, opt_code_dict=codes, pred_dict={"coarse_dict": pred}, gt_rgb=gt,
                                 face_mask_tensor=face, full_eye_mask_tensor=full_eye, left_eye_mask_tensor=leye,
                                 right_eye_mask_tensor=reye, cam_ind=None, ldms=None, epoch=0, batch_num=0)
        ours = L.total_loss(pred, gt, face, full_eye, leye, reye, codes, delta, use_l1=use_l1)
        for k, v in rl.items():
            e = abs(float(v) - float(ours[k]))
            print("  %s %-14s ref %.8f ours %.8f" % ("l1" if use_l1 else "l2", k, float(v), float(ours[k])))
            assert e <= 1e-6 * max(1.0, abs(float(v))), k
            arrays[("l1_" if use_l1 else "l2_") + k] = np.float64(float(v))
    ang = 0.7 * torch.randn(B, 3, generator=g)
    holder = types.SimpleNamespace(batch_size=B)
    rref = BaseTrainer.eulurangle2Rmat(holder, ang)
    e = float((rref - L.euler_to_rotation(ang)).abs().max())
    print("  eulurangle2Rmat max-abs %.2e" % e)
    assert e <= 1e-6
    arrays["euler"], arrays["euler_R"] = ang, rref
    np.savez_compressed(os.path.join(GOLD, "g10_losses.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print("  wrote tests/golden/g10_losses.npz")


if __name__ == "__main__":
    main()
