"""Dataset-sample layout and the sample -> op-input step of the reference's training loop.

What the reference does between its DataLoader and the network:

* ``ETHXGazeDataset.__getitem__`` (datasets/eth_xgaze.py:308-433) reads one row of an ``xgaze_subjectXXXX.h5`` file --
  the field / dtype contract is written down in :class:`XGazeRow` (dataset_pre_processing.py:260-381 creates the
  datasets) -- and turns it into ``(image, head_mask, left_eye_mask, right_eye_mask, nl3dmm_para_dict)``:
  BGR -> RGB, ``ToPILImage`` + ``ToTensor`` (u8 HWC -> float CHW in [0,1]), the head mask eroded twice with a 3x3
  kernel, and the 306-float latent code assembled from row 0 of the file (identity / expression / texture are
  per-subject) with only the 27 illumination entries ``[279:]`` from the row itself (:346-347).
  -> :func:`row_to_sample`.  The HDF5 reader itself needs h5py, which is not available offline; any mapping of field
  name -> numpy array (one row) is accepted.
* the DataLoader stacks samples -> :func:`collate`.
* ``GazeNerfTrainer.prepare_data`` (trainer/gazenerf_trainer.py:250-336) splits the code 100 / 79 / 100 / 27,
  OVERWRITES the expression part with one fixed expression code (:304-310, ``configs/config_files/tensor.pt``; the
  caller passes that tensor, it is data of the reference and is not shipped here), moves the camera to fp32 and rescales
  the intrinsics to the feature-map resolution (``inmat[:, :2, :] *= featmap_size / img_size``, :318) before the
  closed-form inverse (:320-325).  -> :func:`prepare_batch`, whose ``base`` dict is what ``losses.Fitter`` consumes.

Plain host-side PyTorch: a few hundred bytes per sample, nothing here is a hot path.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

IDEN_DIMS, EXPR_DIMS, TEXT_DIMS, ILLU_DIMS = 100, 79, 100, 27          # configs/gazenerf_options.py:12-15
CODE_DIMS = IDEN_DIMS + EXPR_DIMS + TEXT_DIMS + ILLU_DIMS               # 306
ILLU_START = IDEN_DIMS + EXPR_DIMS + TEXT_DIMS                          # 279: the per-row part of latent_codes


@dataclass
class XGazeRow:
    """One row of the reference's pre-processed ETH-XGaze HDF5 file (dataset_pre_processing.py:260-381; read at
    datasets/eth_xgaze.py:326-352).  ``np.float`` there is float64.  Shapes are per row."""
    face_patch: np.ndarray        # [512,512,3] uint8, BGR
    head_mask: np.ndarray         # [512,512]   uint8 (0 / 1)
    left_eye_mask: np.ndarray     # [512,512]   uint8
    right_eye_mask: np.ndarray    # [512,512]   uint8
    latent_codes: np.ndarray      # [306] float64: iden 100 | expr 79 | text 100 | illu 27
    w2c_Rmat: np.ndarray          # [3,3] float64
    w2c_Tvec: np.ndarray          # [3]   float64
    c2w_Rmat: np.ndarray          # [3,3] float64
    c2w_Tvec: np.ndarray          # [3]   float64
    inmat: np.ndarray             # [3,3] float64, intrinsics at the image resolution (512)
    inv_inmat: np.ndarray         # [3,3] float64
    pitchyaw_head: np.ndarray     # [2]   float64, gaze (pitch, yaw) in the head coordinate system
    face_head_pose: np.ndarray    # [2]   float64
    facial_landmarks: Optional[np.ndarray] = None    # [68,2] float64
    cam_index: Optional[np.ndarray] = None           # [1] uint8

    SHAPES = {"face_patch": (None, None, 3), "head_mask": (None, None), "left_eye_mask": (None, None),
              "right_eye_mask": (None, None), "latent_codes": (CODE_DIMS,), "w2c_Rmat": (3, 3), "w2c_Tvec": (3,),
              "c2w_Rmat": (3, 3), "c2w_Tvec": (3,), "inmat": (3, 3), "inv_inmat": (3, 3), "pitchyaw_head": (2,),
              "face_head_pose": (2,)}
    U8 = ("face_patch", "head_mask", "left_eye_mask", "right_eye_mask")

    def validate(self):
        for name, shape in self.SHAPES.items():
            a = np.asarray(getattr(self, name))
            if a.ndim != len(shape) or any(s is not None and s != d for s, d in zip(shape, a.shape)):
                raise ValueError("XGazeRow.%s must have shape %s, got %s" % (name, shape, a.shape))
            if name in self.U8 and a.dtype != np.uint8:
                raise ValueError("XGazeRow.%s must be uint8, got %s" % (name, a.dtype))
            if name not in self.U8 and a.dtype.kind != "f":
                raise ValueError("XGazeRow.%s must be floating point, got %s" % (name, a.dtype))
        if self.head_mask.shape != self.face_patch.shape[:2]:
            raise ValueError("XGazeRow: masks and face_patch must share their resolution")
        return self

    @classmethod
    def from_mapping(cls, m: Mapping[str, np.ndarray]):
        names = [f.name for f in fields(cls)]
        return cls(**{k: np.asarray(m[k]) for k in names if k in m}).validate()


def erode3x3(mask: torch.Tensor, iterations: int = 2) -> torch.Tensor:
    """cv2.erode(mask, ones(3,3), iterations=n) (datasets/eth_xgaze.py:338-339): a 3x3 minimum filter, n times;
    cv2's default border value for erosion is +inf, i.e. pixels outside never lower the minimum -- max_pool2d pads
    with -inf, the same statement for max(-x).  cv2 is not available offline: restated from its documentation,
    UNPINNED (like the kornia blur of the upsampler)."""
    x = mask.to(torch.float32)[None, None] if mask.dim() == 2 else mask.to(torch.float32)
    for _ in range(iterations):
        x = -F.max_pool2d(-x, 3, stride=1, padding=1)
    return (x[0, 0] if mask.dim() == 2 else x).to(mask.dtype)


def row_to_sample(row: XGazeRow, subject_row0_codes: np.ndarray):
    """datasets/eth_xgaze.py:326-352 for one row: -> (image [3,H,W] f32 RGB in [0,1], head_mask [H,W] u8 (eroded),
    left_eye_mask, right_eye_mask, nl3dmm_para_dict).  ``subject_row0_codes`` = ``latent_codes[0]`` of the same file."""
    row.validate()
    img = torch.from_numpy(np.ascontiguousarray(row.face_patch[:, :, [2, 1, 0]]))           # BGR -> RGB
    img = img.permute(2, 0, 1).to(torch.float32) / 255.0                                    # ToPILImage + ToTensor
    code = np.array(subject_row0_codes, dtype=np.float64, copy=True)
    code[ILLU_START:] = row.latent_codes[ILLU_START:]
    para = {"code": code, "w2c_Rmat": row.w2c_Rmat, "w2c_Tvec": row.w2c_Tvec, "inmat": row.inmat,
            "c2w_Rmat": row.c2w_Rmat, "c2w_Tvec": row.c2w_Tvec, "inv_inmat": row.inv_inmat,
            "pitchyaw": row.pitchyaw_head, "head_pose": row.face_head_pose, "eye_mask": 0}
    head = erode3x3(torch.from_numpy(np.ascontiguousarray(row.head_mask)), 2)
    return (img, head, torch.from_numpy(np.ascontiguousarray(row.left_eye_mask)),
            torch.from_numpy(np.ascontiguousarray(row.right_eye_mask)), para)


def collate(samples: List[tuple]):
    """torch's default_collate for the 5-tuples of row_to_sample (numpy -> tensor, stacked along a new batch axis)."""
    imgs, heads, lefts, rights, paras = zip(*samples)
    para = {k: torch.stack([torch.as_tensor(np.asarray(p[k])) for p in paras]) for k in paras[0]}
    return torch.stack(imgs), torch.stack(heads), torch.stack(lefts), torch.stack(rights), para


@dataclass
class PreparedBatch:
    """What ``GazeNerfTrainer.prepare_data`` leaves on ``self`` (trainer/gazenerf_trainer.py:250-336), as values."""
    img: torch.Tensor                 # [B,3,H,W] f32
    head_mask: torch.Tensor           # [B,1,H,W]
    left_eye_mask: torch.Tensor       # [B,1,H,W]
    right_eye_mask: torch.Tensor      # [B,1,H,W]
    full_eye_mask: torch.Tensor       # [B,1,1,1]: the dataset's constant ``eye_mask`` 0 per row (eth_xgaze.py:352)
    base: Dict[str, torch.Tensor] = field(default_factory=dict)
    # base: iden [B,100] | expr [B,79] (the FIXED expression) | text [B,100] | illu [B,27] | gaze [B,2] |
    #       c2w_Rmat [B,3,3] | c2w_Tvec [B,3,1] | inv_inmat [B,3,3] (feature-map scale) | inmat [B,3,3] (scaled) |
    #       w2c_Rmat [B,3,3] | w2c_Tvec [B,3,1]


def prepare_batch(img, head_mask, left_eye_mask, right_eye_mask, para: Mapping[str, torch.Tensor], *,
                  base_expr_fix: torch.Tensor, featmap_size: int = 64, pred_img_size: int = 512,
                  device=None) -> PreparedBatch:
    """``GazeNerfTrainer.prepare_data`` (trainer/gazenerf_trainer.py:250-336).

    ``para``: the collated ``nl3dmm_para_dict`` -- code [B,306] f64, pitchyaw [B,2], c2w_Rmat / w2c_Rmat [B,3,3],
    c2w_Tvec / w2c_Tvec [B,3], inmat [B,3,3] (+ eye_mask).  ``base_expr_fix`` [1,79]: the fixed expression code the
    reference loads from ``configs/config_files/tensor.pt`` and substitutes for every sample's own expression
    (:304-310).  The intrinsics are rescaled in float64 and inverted in closed form BEFORE the cast to fp32, as the
    reference does (:317-334)."""
    dev = torch.device(device) if device is not None else img.device
    B = img.shape[0]
    code = torch.as_tensor(para["code"]).detach()
    if code.dim() != 2 or code.shape != (B, CODE_DIMS):
        raise ValueError("para['code'] must be [%d,%d], got %s" % (B, CODE_DIMS, tuple(code.shape)))
    f32 = lambda t: torch.as_tensor(t).detach().to(torch.float32).to(dev)
    if tuple(base_expr_fix.shape) != (1, EXPR_DIMS):
        raise ValueError("base_expr_fix must be [1,%d], got %s" % (EXPR_DIMS, tuple(base_expr_fix.shape)))
    base = {
        "iden": f32(code[:, :IDEN_DIMS]),
        "expr": f32(base_expr_fix.expand(B, -1)),                       # the sample's own code[:, 100:179] is discarded
        "text": f32(code[:, IDEN_DIMS + EXPR_DIMS:ILLU_START]),
        "illu": f32(code[:, ILLU_START:]),
        "gaze": f32(para["pitchyaw"]),
        "c2w_Rmat": f32(para["c2w_Rmat"]),
        "c2w_Tvec": f32(torch.as_tensor(para["c2w_Tvec"]).unsqueeze(-1)),
    }
    if "w2c_Rmat" in para:
        base["w2c_Rmat"] = f32(para["w2c_Rmat"])
        base["w2c_Tvec"] = f32(torch.as_tensor(para["w2c_Tvec"]).unsqueeze(-1))
    inmat = torch.as_tensor(para["inmat"]).detach().clone()
    inmat[:, :2, :] *= featmap_size / pred_img_size
    inv = torch.zeros_like(inmat)
    inv[:, 0, 0] = 1.0 / inmat[:, 0, 0]
    inv[:, 1, 1] = 1.0 / inmat[:, 1, 1]
    inv[:, 0, 2] = -(inmat[:, 0, 2] / inmat[:, 0, 0])
    inv[:, 1, 2] = -(inmat[:, 1, 2] / inmat[:, 1, 1])
    inv[:, 2, 2] = 1.0
    base["inmat"], base["inv_inmat"] = f32(inmat), f32(inv)
    m = lambda t: torch.as_tensor(t).clone().unsqueeze(1).to(dev)
    eye = torch.as_tensor(para.get("eye_mask", torch.zeros(B))).to(torch.float32).reshape(B, 1, 1, 1).to(dev)
    return PreparedBatch(img=torch.as_tensor(img).clone().to(dev), head_mask=m(head_mask), left_eye_mask=m(left_eye_mask),
                         right_eye_mask=m(right_eye_mask), full_eye_mask=eye, base=base)
