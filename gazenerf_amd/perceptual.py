"""The VGG-perceptual term of the reference's training loss: module structure and loss arithmetic.

Restated from losses/gazenerf_loss.py:40-102 (``VGGPerceptualLoss``) and its three uses in ``calc_data_loss`` (:360-381).
The reference builds its feature extractor from ``torchvision.models.vgg16(pretrained=True).features[:23]``; neither
torchvision nor the ImageNet weights exist offline, so

  * ``vgg16_features()`` restates the LAYOUT of torchvision's ``vgg16().features`` up to index 22 (configuration "D":
    3x3 convolutions 64 64 M 128 128 M 256 256 256 M 512 512 512, ReLU after each, 2x2 max-pools at indices 4, 9, 16),
    with the same module indices, so a torchvision checkpoint (``features.<i>.weight / bias``) loads with
    ``load_torchvision_vgg16`` -- the architecture is UNPINNED against torchvision itself (it is not importable here);
  * the loss ARITHMETIC (ImageNet normalisation, bilinear resize to 224, the four slices [:4] [4:9] [9:16] [16:23], L1
    between feature maps, optional Gram terms, and how ``calc_data_loss`` masks its targets and weights the three terms)
    is pinned: oracle/gen_golden_vgg.py runs the reference's own ``VGGPerceptualLoss`` / ``GazeNeRFLoss`` with
    ``torchvision.models.vgg16`` replaced by a factory that returns this layout with hashed weights (fixture g14).

Without weights the module is randomly initialised and says so (``pretrained`` is False): a perceptual loss on random
features is a valid regulariser of the code path, not the reference's loss.  Plain PyTorch-ROCm, outside the hot path.

The gaze-angular term (``GazePerceptualLoss``, gazenerf_loss.py:105-190; ``gaze_network``,
gaze_estimation/xgaze_baseline_vgg.py:6-46) is here on the same terms: ``GazeNetwork`` = the full ``vgg16().features`` (31
modules) + the three-layer head under the reference's parameter names (its checkpoint's ``model_state`` loads),
``GazeAngularLoss.forward`` = normalise, resize, estimate pitch / yaw of both images, mean angle in degrees between the two gaze
vectors (target detached); ``total_loss(gaze=...)`` adds ``angular / 60000 * eye_loss_importance`` (:389-391).  The reference's
constructor also reads camera files through cv2 that its forward never uses; nothing of that is needed.  Arithmetic pinned by
running the reference's own ``forward`` and ``gaze_network`` on hashed weights (fixture g14); the estimator's trained weights
(``epoch_60_512_ckpt.pth.tar``) are the caller's to supply.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Iterable, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_VGG16_D_TO_22 = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512)        # features[0:23]
_VGG16_D = _VGG16_D_TO_22 + ("M", 512, 512, 512, "M")                                    # features[0:31]
_SLICES = ((0, 4), (4, 9), (9, 16), (16, 23))                                           # gazenerf_loss.py:50-53


def _make_features(cfg) -> nn.Sequential:
    layers, cin = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def vgg16_features() -> nn.Sequential:
    """torchvision ``vgg16().features[:23]``: index 0 conv(3,64) 1 relu 2 conv 3 relu 4 pool 5 conv(64,128) ... 21 conv(512,512) 22 relu."""
    f = _make_features(_VGG16_D_TO_22)
    assert len(f) == 23
    return f


def vgg16_features_full() -> nn.Sequential:
    """torchvision ``vgg16().features`` complete: the 23 modules above, then 23 pool, 24 / 26 / 28 conv(512,512) + ReLUs, 30 pool."""
    f = _make_features(_VGG16_D)
    assert len(f) == 31
    return f


def load_torchvision_vgg16(features: nn.Sequential, state_dict: Dict[str, torch.Tensor]) -> None:
    """Fill ``features`` from a torchvision VGG-16 checkpoint (keys ``features.<i>.weight`` / ``.bias``; classifier entries and
    feature layers beyond index 22 are ignored).  Every convolution of the layout must be present."""
    own = features.state_dict()
    picked = OrderedDict()
    for k in own:
        src = "features." + k
        if src not in state_dict:
            raise KeyError("load_torchvision_vgg16: %r is missing from the checkpoint" % src)
        if tuple(state_dict[src].shape) != tuple(own[k].shape):
            raise ValueError("load_torchvision_vgg16: %s has shape %s, expected %s" % (src, tuple(state_dict[src].shape), tuple(own[k].shape)))
        picked[k] = state_dict[src]
    features.load_state_dict(picked, strict=True)


class VGGPerceptualLoss(nn.Module):
    """gazenerf_loss.py:40-102.  ``blocks`` = the four slices of the feature extractor (frozen, eval mode);
    ``forward(input, target)`` = sum over the chosen blocks of L1(feature(input), feature(target)) (+ L1 of Gram matrices for
    ``style_layers``) after ImageNet normalisation and a bilinear resize to 224 x 224."""

    def __init__(self, resize: bool = True, features: Optional[nn.Sequential] = None, pretrained_state: Optional[Dict[str, torch.Tensor]] = None):
        super().__init__()
        feats = features if features is not None else vgg16_features()
        if pretrained_state is not None:
            load_torchvision_vgg16(feats, pretrained_state)
        self.pretrained = pretrained_state is not None or features is not None
        self.blocks = nn.ModuleList([feats[a:b].eval() for a, b in _SLICES])
        for q in self.blocks.parameters():
            q.requires_grad = False
        self.resize = resize
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def train(self, mode: bool = True):          # the extractor stays in eval mode (it has no mode-dependent layer; kept explicit)
        return super().train(False)

    def forward(self, input, target, feature_layers: Iterable[int] = (0, 1, 2, 3), style_layers: Iterable[int] = ()):
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        input, target = (input - self.mean) / self.std, (target - self.mean) / self.std
        if self.resize:
            input = F.interpolate(input, mode="bilinear", size=(224, 224), align_corners=False)
            target = F.interpolate(target, mode="bilinear", size=(224, 224), align_corners=False)
        loss = 0.0
        x, y = input, target
        for i, block in enumerate(self.blocks):
            x, y = block(x), block(y)
            if i in feature_layers:
                loss = loss + F.l1_loss(x, y)
            if i in style_layers:
                ax, ay = x.reshape(x.shape[0], x.shape[1], -1), y.reshape(y.shape[0], y.shape[1], -1)
                loss = loss + F.l1_loss(ax @ ax.permute(0, 2, 1), ay @ ay.permute(0, 2, 1))
        return loss

    def forward_groups(self, input, target, groups: int, feature_layers: Iterable[int] = (0, 1, 2, 3)):
        """``forward`` for ``groups`` stacked (input, target) batches at once: ``[forward(input[g], target[g]) for g]`` with
        one pass of the extractor over all inputs and one (without autograd) over all targets."""
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        input, target = (input - self.mean) / self.std, (target - self.mean) / self.std
        if self.resize:
            input = F.interpolate(input, mode="bilinear", size=(224, 224), align_corners=False)
            target = F.interpolate(target, mode="bilinear", size=(224, 224), align_corners=False)
        n = input.shape[0] // groups
        losses = [0.0] * groups
        x, y = input, target
        for i, block in enumerate(self.blocks):
            x = block(x)
            with torch.no_grad():
                y = block(y)
            if i in feature_layers:
                for g in range(groups):
                    losses[g] = losses[g] + F.l1_loss(x[g * n:(g + 1) * n], y[g * n:(g + 1) * n])
        return losses


def vgg_terms(vgg: VGGPerceptualLoss, pred: Dict[str, torch.Tensor], gt_rgb, masks, bg_value: float, vgg_importance: float):
    """gazenerf_loss.py:360-381: the face / eyes predictions against the ground truth with everything outside the region painted
    in the background colour, and the merged image against the ground truth with the non-head region painted; only the last
    one carries ``vgg_importance``.

    The reference calls its module three times = six passes of B images through the extractor.  Here the three predictions go
    through it as ONE batch of 3B and the three targets as another (under ``no_grad``: they carry no gradient); per term the same
    four L1 means are added in the same order, so the values are the reference's (the convolutions are per-sample) at a third
    of the launches -- on the MI355X the three terms cost 8.7 ms per B = 2 step the reference's way (profiles/r4_s22_*)."""
    c3 = lambda m: m.expand(-1, 3, -1, -1)
    bg = torch.full_like(gt_rgb, bg_value)
    B = gt_rgb.shape[0]
    inputs = torch.cat([pred["merge_img_face"], pred["merge_img_eyes"], pred["merge_img"]], dim=0)
    targets = torch.cat([torch.where(c3(masks["face"]), gt_rgb, bg), torch.where(c3(masks["eyes"]), gt_rgb, bg),
                         torch.where(c3(masks["nonhead"]), bg, gt_rgb)], dim=0)
    per_term = vgg.forward_groups(inputs, targets, groups=3)
    assert inputs.shape[0] == 3 * B
    out = OrderedDict()
    out["vgg_face_loss"], out["vgg_eyes_loss"] = per_term[0], per_term[1]
    out["vgg"] = per_term[2] * vgg_importance
    return out


class GazeNetwork(nn.Module):
    """gaze_estimation/xgaze_baseline_vgg.py:6-46: VGG-16 features, global average pool, FC 512-64-64-4 with LeakyReLU(0.2),
    tanh, x pi/2 -> (gaze pitch / yaw, head pitch / yaw).  Parameter names are the reference's (``vgg16.<i>``, ``FC1`` ...)."""

    def __init__(self, features: Optional[nn.Sequential] = None):
        super().__init__()
        self.vgg16 = features if features is not None else vgg16_features_full()
        self.FC1, self.FC2, self.FC3 = nn.Linear(512, 64), nn.Linear(64, 64), nn.Linear(64, 4)
        self.act, self.tanh = nn.LeakyReLU(0.2, True), nn.Tanh()
        for fc in (self.FC1, self.FC2, self.FC3):
            nn.init.kaiming_normal_(fc.weight.data)
            nn.init.constant_(fc.bias.data, val=0)

    def forward(self, x):
        h = self.vgg16(x).mean(-1).mean(-1)
        h = self.act(self.FC1(h))
        h = self.act(self.FC2(h))
        h = math.pi * 0.5 * self.tanh(self.FC3(h))
        return h[:, :2], h[:, 2:]


def pitchyaw_to_vector(pitchyaws: torch.Tensor) -> torch.Tensor:
    """gazenerf_loss.py:147-150."""
    sin, cos = torch.sin(pitchyaws), torch.cos(pitchyaws)
    return torch.stack([cos[:, 0] * sin[:, 1], sin[:, 0], cos[:, 0] * cos[:, 1]], 1)


def angular_distance_deg(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """gazenerf_loss.py:142-145: per-row angle in degrees (cosine similarity with eps 1e-6, clamped to [-1, 1])."""
    sim = F.hardtanh(F.cosine_similarity(a, b, eps=1e-6), -1.0, 1.0)
    return torch.acos(sim) * (180 / math.pi)


class GazeAngularLoss(nn.Module):
    """gazenerf_loss.py:105-190 (``GazePerceptualLoss``): mean angle between the gaze the estimator reads off the generated image
    and off the (detached) target.  ``model``: a ``GazeNetwork`` -- with ``model_state`` the reference checkpoint's
    ``model_state`` entry is loaded into it."""

    def __init__(self, model: Optional[GazeNetwork] = None, model_state: Optional[Dict[str, torch.Tensor]] = None, resize: bool = True):
        super().__init__()
        self.model = model if model is not None else GazeNetwork()
        if model_state is not None:
            self.model.load_state_dict(model_state, strict=True)
        self.pretrained = model is not None or model_state is not None
        self.model.eval()
        self.resize = resize
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def train(self, mode: bool = True):
        return super().train(False)

    def forward(self, input, target):
        if input.shape[1] != 3:
            input, target = input.repeat(1, 3, 1, 1), target.repeat(1, 3, 1, 1)
        input, target = (input - self.mean) / self.std, (target - self.mean) / self.std
        if self.resize:         # trans_eval: torchvision Resize((224, 224)) on a tensor == bilinear, no antialias (UNPINNED, see gan.resize_224)
            input = F.interpolate(input, size=(224, 224), mode="bilinear", align_corners=False)
            target = F.interpolate(target, size=(224, 224), mode="bilinear", align_corners=False)
        gaze_x, _ = self.model(input)
        gaze_y, _ = self.model(target)
        return torch.mean(angular_distance_deg(pitchyaw_to_vector(gaze_y.detach()), pitchyaw_to_vector(gaze_x)))


def angular_term(gaze: GazeAngularLoss, pred: Dict[str, torch.Tensor], gt_rgb, masks, bg_value: float, eye_loss_importance: float):
    """gazenerf_loss.py:383-391: merged image against the ground truth with the non-head region painted, / 60000, x the eye weight."""
    bg = torch.full_like(gt_rgb, bg_value)
    target = torch.where(masks["nonhead"].expand(-1, 3, -1, -1), bg, gt_rgb)
    return (gaze(pred["merge_img"], target) / 60000.0) * eye_loss_importance


# ---------------------------------------------------------------------------------------------------------------
# deterministic feature weights for the fixture and the tests (He-uniform from the counter-based hash of gazenerf_amd.synth:
# activations keep their scale through the 10 convolutions, so every block contributes to the loss)
# ---------------------------------------------------------------------------------------------------------------
def hash_vgg16_state(seed: int = 0, full: bool = False) -> "OrderedDict[str, torch.Tensor]":
    from .synth import _key, hash_uniform
    out = OrderedDict()
    for k, v in (vgg16_features_full() if full else vgg16_features()).state_dict().items():
        if k.endswith("weight"):
            cout, cin = v.shape[0], v.shape[1]
            bound = math.sqrt(6.0 / (cin * 9))
            u = hash_uniform(v.numel(), _key("vgg16." + k, seed))
            out[k] = torch.from_numpy(((2.0 * u - 1.0) * bound).astype(np.float32).reshape(v.shape))
        else:
            u = hash_uniform(v.numel(), _key("vgg16." + k, seed))
            out[k] = torch.from_numpy((0.1 * (u - 0.5)).astype(np.float32))
    return out


def hash_gaze_head_state(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """FC1 / FC2 / FC3 of the gaze estimator from the hash (He-uniform weights, small biases)."""
    from .synth import _key, hash_uniform
    out = OrderedDict()
    for name, (cout, cin) in (("FC1", (64, 512)), ("FC2", (64, 64)), ("FC3", (4, 64))):
        bound = math.sqrt(6.0 / cin)
        u = hash_uniform(cout * cin, _key("gaze." + name + ".weight", seed))
        out[name + ".weight"] = torch.from_numpy(((2.0 * u - 1.0) * bound).astype(np.float32).reshape(cout, cin))
        ub = hash_uniform(cout, _key("gaze." + name + ".bias", seed))
        out[name + ".bias"] = torch.from_numpy((0.2 * (ub - 0.5)).astype(np.float32))
    return out
