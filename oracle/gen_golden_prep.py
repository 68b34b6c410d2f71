"""ORACLE tooling -- test infrastructure, NOT product code.

Fixture for SURVEY.md 8(f) N4, the sample -> op-input step: the reference's own ``GazeNerfTrainer.prepare_data``
and ``GazeNerfTrainer.build_code_and_cam`` (trainer/gazenerf_trainer.py:250-405) run here, as unbound functions on a
plain namespace carrying exactly the attributes they read (the trainer's constructor builds the network, reads files
and talks to wandb; none of that is on this path), on a seeded synthetic dataset batch with the field / dtype
contract of datasets/eth_xgaze.py:326-352.  torchvision / cv2 / wandb ... are import-time dependencies only and are
replaced by empty modules.  Asserts gazenerf_amd.data.prepare_batch + losses.Fitter.build_code_and_cam == reference,
writes tests/golden/g12_prepare.npz (inputs incl. the reference's fixed expression code, a 79-float data tensor, and
every output).

    python oracle/gen_golden_prep.py
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


def synthetic_batch(B=3, S=32, seed=11):
    """A DataLoader batch with the dataset's dtypes: image f32 [B,3,S,S], masks u8 [B,S,S], para dict f64."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, S, S, generator=g)
    mk = lambda p: (torch.rand(B, S, S, generator=g) < p).to(torch.uint8)
    code = (0.6 * torch.randn(B, 306, generator=g)).double()
    ang = 0.3 * torch.randn(B, 3, generator=g).double()
    c, s = torch.cos(ang), torch.sin(ang)
    R = torch.zeros(B, 3, 3, dtype=torch.float64)
    for b in range(B):                                  # some proper rotation per row
        rz = torch.tensor([[c[b, 2], -s[b, 2], 0], [s[b, 2], c[b, 2], 0], [0, 0, 1]], dtype=torch.float64)
        ry = torch.tensor([[c[b, 1], 0, s[b, 1]], [0, 1, 0], [-s[b, 1], 0, c[b, 1]]], dtype=torch.float64)
        R[b] = rz @ ry @ torch.diag(torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64))
    T = torch.tensor([0.0, 0.0, 12.0], dtype=torch.float64) + 0.4 * torch.randn(B, 3, generator=g).double()
    inmat = torch.zeros(B, 3, 3, dtype=torch.float64)
    inmat[:, 0, 0] = 2050.0 + 30 * torch.rand(B, generator=g).double()
    inmat[:, 1, 1] = 2052.0 + 30 * torch.rand(B, generator=g).double()
    inmat[:, 0, 2] = 256.0 + 5 * torch.randn(B, generator=g).double()
    inmat[:, 1, 2] = 262.0 + 5 * torch.randn(B, generator=g).double()
    inmat[:, 2, 2] = 1.0
    para = {"code": code, "pitchyaw": 0.4 * torch.randn(B, 2, generator=g).double(), "c2w_Rmat": R, "c2w_Tvec": T,
            "w2c_Rmat": R.transpose(1, 2).contiguous(), "w2c_Tvec": -(R.transpose(1, 2) @ T.unsqueeze(-1)).squeeze(-1),
            "inmat": inmat, "inv_inmat": torch.linalg.inv(inmat), "head_pose": torch.zeros(B, 2, dtype=torch.float64),
            "eye_mask": torch.zeros(B, dtype=torch.int64)}
    return img, mk(0.6), mk(0.1), mk(0.1), para


def main():
    sys.dont_write_bytecode = True
    for name in ("cv2", "torchvision", "gaze_estimation", "gaze_estimation.xgaze_baseline_vgg", "wandb", "imageio",
                 "skimage", "skimage.metrics", "piq", "kornia", "kornia.filters", "h5py", "lpips", "face_recognition",
                 "tqdm", "PIL", "PIL.Image"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    sys.modules["kornia.filters"].filter2d = None
    tv = sys.modules["torchvision"]
    tr = types.ModuleType("torchvision.transforms")
    ident = lambda *a, **k: (lambda x: x)
    tr.Compose, tr.ToPILImage, tr.ToTensor, tr.Normalize, tr.Resize = (lambda fs: (lambda x: x)), ident, ident, ident, ident
    tv.transforms = tr
    tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision.transforms"] = tr
    sys.modules["torchvision.models"] = tv.models
    sys.modules["gaze_estimation.xgaze_baseline_vgg"].gaze_network = object
    sys.path.insert(0, REF)
    os.chdir(REF)
    from configs.gazenerf_options import BaseOptions
    from trainer.base import BaseTrainer
    from trainer.gazenerf_trainer import GazeNerfTrainer

    from gazenerf_amd import data as D
    from gazenerf_amd import losses as L

    B, S, FM = 2, 64, 16                      # pred_img_size 64, featmap_size 16 (the reference: 512 and 64)
    img, head, leye, reye, para = synthetic_batch(B, S)
    expr_fix = torch.load(os.path.join(REF, "configs", "config_files", "tensor.pt"))
    assert tuple(expr_fix.shape) == (1, 79) and expr_fix.dtype == torch.float32

    # ---- the reference, as unbound functions on a namespace ----
    me = types.SimpleNamespace(opt=BaseOptions(), device=torch.device("cpu"), pred_img_size=S, featmap_size=FM,
                               batch_size=B, base_expr_fix=expr_fix, opt_cam=True)
    GazeNerfTrainer.prepare_data(me, img, head, leye, reye, {k: v.clone() for k, v in para.items()})
    g = torch.Generator().manual_seed(5)
    n_rows = 7
    me.iden_offset = 0.1 * torch.randn(n_rows, 100, generator=g)
    me.expr_offset = 0.1 * torch.randn(n_rows, 79, generator=g)
    me.appea_offset = 0.1 * torch.randn(n_rows, 127, generator=g)
    me.delta_EulurAngles = 0.05 * torch.randn(n_rows, 3, generator=g)
    me.delta_Tvecs = 0.05 * torch.randn(n_rows, 3, 1, generator=g)
    me.eulurangle2Rmat = types.MethodType(BaseTrainer.eulurangle2Rmat, me)
    it = 1                                      # rows 2..3 of the offset tables
    code_info, opt_code, cam_info, delta_info = GazeNerfTrainer.build_code_and_cam(me, it)

    # ---- ours ----
    pb = D.prepare_batch(img, head, leye, reye, para, base_expr_fix=expr_fix, featmap_size=FM, pred_img_size=S)
    ref_base = {"iden": me.base_iden, "expr": me.base_expr, "text": me.base_text, "illu": me.base_illu,
                "gaze": me.base_gaze_direction, "c2w_Rmat": me.cam_info["batch_Rmats"],
                "c2w_Tvec": me.cam_info["batch_Tvecs"], "inv_inmat": me.cam_info["batch_inv_inmats"],
                "inmat": me.temp_inmat.float()}
    for k, v in ref_base.items():
        assert pb.base[k].dtype == torch.float32 and torch.equal(pb.base[k], v), k
    assert torch.equal(pb.img, me.img_tensor) and torch.equal(pb.head_mask, me.head_mask_tensor)
    assert torch.equal(pb.left_eye_mask, me.left_eye_mask_tensor) and torch.equal(pb.right_eye_mask, me.right_eye_mask_tensor)
    fit = L.Fitter.__new__(L.Fitter)
    fit.opt_cam = True
    fit.iden_offset, fit.expr_offset, fit.appea_offset = me.iden_offset, me.expr_offset, me.appea_offset
    fit.delta_EulurAngles, fit.delta_Tvecs = me.delta_EulurAngles, me.delta_Tvecs
    rows = slice(it * B, (it + 1) * B)
    shape_code, appea_code, gaze, R, T, opt_codes, delta = fit.build_code_and_cam(rows, pb.base)
    outs = {"shape_code": (shape_code, code_info["shape_code"]), "appea_code": (appea_code, code_info["appea_code"]),
            "gaze_code": (gaze, code_info["gaze_code"]), "R": (R, cam_info["batch_Rmats"]), "T": (T, cam_info["batch_Tvecs"]),
            "inv_inmat": (pb.base["inv_inmat"], cam_info["batch_inv_inmats"])}
    for k, (a, b) in outs.items():
        e = float((a - b).abs().max())
        print("  %-11s max-abs vs reference %.2e" % (k, e))
        assert e <= 1e-6, k
    arrays = {"img": img, "head": head, "leye": leye, "reye": reye, "expr_fix": expr_fix,
              **{"para_" + k: v for k, v in para.items()},
              **{"base_" + k: v for k, v in ref_base.items()},
              "iden_offset": me.iden_offset, "expr_offset": me.expr_offset, "appea_offset": me.appea_offset,
              "delta_EulurAngles": me.delta_EulurAngles, "delta_Tvecs": me.delta_Tvecs,
              **{"out_" + k: b for k, (a, b) in outs.items()},
              "meta": np.array([B, S, FM, it], dtype=np.int64)}
    np.savez_compressed(os.path.join(GOLD, "g12_prepare.npz"), **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print("  wrote tests/golden/g12_prepare.npz")


if __name__ == "__main__":
    main()
