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
_SLICES = ((0, 4), (4, 9), (9, 16), (16, 23))                                           # gazenerf_loss.py:50-53


def vgg16_features() -> nn.Sequential:
    """torchvision ``vgg16().features[:23]``: index 0 conv(3,64) 1 relu 2 conv 3 relu 4 pool 5 conv(64,128) ... 21 conv(512,512) 22 relu."""
    layers, cin = [], 3
    for v in _VGG16_D_TO_22:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    assert len(layers) == 23
    return nn.Sequential(*layers)


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


def vgg_terms(vgg: VGGPerceptualLoss, pred: Dict[str, torch.Tensor], gt_rgb, masks, bg_value: float, vgg_importance: float):
    """gazenerf_loss.py:360-381: the face / eyes predictions against the ground truth with everything outside the region painted
    in the background colour, and the merged image against the ground truth with the non-head region painted; only the last
    one carries ``vgg_importance``."""
    c3 = lambda m: m.expand(-1, 3, -1, -1)
    bg = torch.full_like(gt_rgb, bg_value)
    out = OrderedDict()
    out["vgg_face_loss"] = vgg(pred["merge_img_face"], torch.where(c3(masks["face"]), gt_rgb, bg))
    out["vgg_eyes_loss"] = vgg(pred["merge_img_eyes"], torch.where(c3(masks["eyes"]), gt_rgb, bg))
    out["vgg"] = vgg(pred["merge_img"], torch.where(c3(masks["nonhead"]), bg, gt_rgb)) * vgg_importance
    return out


# ---------------------------------------------------------------------------------------------------------------
# deterministic feature weights for the fixture and the tests (He-uniform from the counter-based hash of gazenerf_amd.synth:
# activations keep their scale through the 10 convolutions, so every block contributes to the loss)
# ---------------------------------------------------------------------------------------------------------------
def hash_vgg16_state(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    from .synth import _key, hash_uniform
    out = OrderedDict()
    for k, v in vgg16_features().state_dict().items():
        if k.endswith("weight"):
            cout, cin = v.shape[0], v.shape[1]
            bound = math.sqrt(6.0 / (cin * 9))
            u = hash_uniform(v.numel(), _key("vgg16." + k, seed))
            out[k] = torch.from_numpy(((2.0 * u - 1.0) * bound).astype(np.float32).reshape(v.shape))
        else:
            u = hash_uniform(v.numel(), _key("vgg16." + k, seed))
            out[k] = torch.from_numpy((0.1 * (u - 0.5)).astype(np.float32))
    return out
