"""The PatchGAN terms of the reference's training step: the discriminator, its loss, the
generator term with its warm-up, and the discriminator update the trainer runs before every generator step.

Restated from models/discriminator.py:4-43 (``PatchGAN``), losses/gazenerf_loss.py:22-38 (``discriminator_loss``,
``generator_loss``), :396-401 (the generator term inside ``calc_data_loss``) and trainer/gazenerf_trainer.py:241-242,
487-508 (the discriminator step).  None of this needs a pretrained network: the discriminator is trained from its
default initialisation beside the renderer.  Plain PyTorch-ROCm (five strided 6x6 convolutions on 224x224 images --
not the hot path); parameter names are the reference's, so a ``PatchGAN.state_dict()`` of either side loads into the
other (tests/test_gan.py, fixture from the reference's own class: oracle/gen_golden_gan.py).

One piece is UNPINNED: ``trans_eval = transforms.Resize((224, 224))`` (gazenerf_loss.py:20).  torchvision is not
available offline; on a float tensor that transform is ``F.interpolate(mode="bilinear", align_corners=False)``
without antialiasing in the torchvision releases that match the reference's torch 1.12 (``antialias=None`` ==
off for tensors) -- restated as such, like the kornia blur and the cv2 erosion.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class PatchGAN(nn.Module):
    """models/discriminator.py:4-43.  kernel 6, padding 1; strides 2, 2, 2, 1, 1; BatchNorm after conv2-4 (those
    convolutions carry no bias); LeakyReLU(0.2); one-channel logit map ([B,1,20,20] for a 224x224 input).

    Gradients through four LeakyReLUs are only piecewise smooth: a pre-activation within rounding distance of zero takes the
    other slope under a different summation order, and ONE such element moves the upstream weight gradients by ~0.7 % (rel-L2)
    -- MIOpen's Winograd backward on the MI355X against oneDNN on the CPU, but equally an im2col + GEMM restatement on the CPU
    against oneDNN (tools/gan_diag.py, profiles/r4_patchgan_gradient_flip.txt).  Not an accuracy defect of either; the fixture
    inputs are chosen with a margin around zero and the GPU test allows for a flip."""

    def __init__(self, input_nc: int = 3, ndf: int = 64):
        super().__init__()
        kw, padw = 6, 1
        self.conv1 = nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw)
        self.conv2 = nn.Conv2d(ndf, ndf * 2, kernel_size=kw, stride=2, padding=padw, bias=False)
        self.norm1 = nn.BatchNorm2d(ndf * 2)
        self.act = nn.LeakyReLU(0.2, True)
        self.conv3 = nn.Conv2d(ndf * 2, ndf * 4, kernel_size=kw, stride=2, padding=padw, bias=False)
        self.norm2 = nn.BatchNorm2d(ndf * 4)
        self.conv4 = nn.Conv2d(ndf * 4, ndf * 8, kernel_size=kw, stride=1, padding=padw, bias=False)
        self.norm3 = nn.BatchNorm2d(ndf * 8)
        self.conv5 = nn.Conv2d(ndf * 8, 1, kernel_size=kw, stride=1, padding=padw)

    def forward(self, x):
        x = self.act(self.conv1(x))
        x = self.act(self.norm1(self.conv2(x)))
        x = self.act(self.norm2(self.conv3(x)))
        x = self.act(self.norm3(self.conv4(x)))
        return self.conv5(x)


def resize_224(img: torch.Tensor) -> torch.Tensor:
    """``trans_eval`` (gazenerf_loss.py:20): torchvision ``Resize((224, 224))`` on a [B,3,H,W] float tensor.  UNPINNED."""
    return F.interpolate(img, size=(224, 224), mode="bilinear", align_corners=False)


def discriminator_loss(real: torch.Tensor, fake: torch.Tensor) -> torch.Tensor:
    """gazenerf_loss.py:22-31.  The reference labels REAL patches 0 and generated ones 1."""
    bce = F.binary_cross_entropy_with_logits
    return (bce(fake, torch.ones_like(fake)) + bce(real, torch.zeros_like(real))) / 2


def generator_loss(fake: torch.Tensor) -> torch.Tensor:
    """gazenerf_loss.py:33-37: the generator wants its patches labelled like real ones (0)."""
    return F.binary_cross_entropy_with_logits(fake, torch.zeros_like(fake))


def warm_up_coeff(epoch: int, batch_num: int) -> float:
    """gazenerf_loss.py:398: ``max(min(1/10, (200000 epoch + batch_num) / 200000), 0)`` -- a ramp over the first 20 000
    batches of epoch 0 that saturates at 0.1 (the weight of the generator term ever after)."""
    return max(min(1.0 / 10.0, (200000 * epoch + batch_num) / 200000), 0.0)


def generator_term(discriminator: nn.Module, merge_img: torch.Tensor, epoch: int, batch_num: int) -> torch.Tensor:
    """gazenerf_loss.py:396-401: ``generator_loss(D(resize(merge_img))) * warm_up``.  Gradients reach the image; the
    caller keeps the discriminator's parameters frozen meanwhile (``DiscriminatorStep`` does, as the trainer)."""
    return generator_loss(discriminator(resize_224(merge_img))) * warm_up_coeff(epoch, batch_num)


class DiscriminatorStep:
    """trainer/gazenerf_trainer.py:110-111, 241-242, 487-508: a PatchGAN, its Adam (the trainer's learning rate,
    weight decay 1e-4) and the update that precedes every generator step --
    real = the ground-truth image with the non-head region painted white, fake = the detached prediction."""

    def __init__(self, device, lr: float = 1e-4, ndf: int = 64):
        self.discriminator = PatchGAN(input_nc=3, ndf=ndf).to(device)
        self.optimizer = torch.optim.Adam(self.discriminator.parameters(), lr=lr, weight_decay=1e-4)

    def _freeze(self, frozen: bool):
        for q in self.discriminator.parameters():
            q.requires_grad = not frozen

    def step(self, gt_rgb: torch.Tensor, face_mask: torch.Tensor, merge_img: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One discriminator update; leaves the parameters frozen for the generator term that follows."""
        self._freeze(False)
        patch_gt = torch.where((face_mask < 0.5).expand(-1, 3, -1, -1), torch.ones_like(gt_rgb), gt_rgb)
        patch_pred = merge_img.detach()
        real = self.discriminator(resize_224(patch_gt))
        fake = self.discriminator(resize_224(patch_pred))
        disc_loss = discriminator_loss(real, fake)
        gen_loss = generator_loss(fake)
        self.optimizer.zero_grad()
        disc_loss.backward()
        self.optimizer.step()
        self._freeze(True)
        return {"disc_loss": disc_loss.detach(), "gen_loss_before_update": gen_loss.detach()}

    def generator_term(self, merge_img: torch.Tensor, epoch: int, batch_num: int) -> torch.Tensor:
        return generator_term(self.discriminator, merge_img, epoch, batch_num)


# ---------------------------------------------------------------------------------------------------------------
# deterministic parameters for fixtures and tests (no shipped weights): Conv2d / BatchNorm default distributions from the
# counter-based hash of gazenerf_amd.synth
# ---------------------------------------------------------------------------------------------------------------
def hash_patchgan_state(seed: int = 0, input_nc: int = 3, ndf: int = 8) -> "OrderedDict[str, torch.Tensor]":
    from .synth import _key, hash_uniform
    out = OrderedDict()
    chans = [(input_nc, ndf, True), (ndf, 2 * ndf, False), (2 * ndf, 4 * ndf, False), (4 * ndf, 8 * ndf, False), (8 * ndf, 1, True)]
    for i, (cin, cout, bias) in enumerate(chans, start=1):
        bound = 1.0 / math.sqrt(cin * 36)
        u = hash_uniform(cout * cin * 36, _key("patchgan.conv%d.weight" % i, seed))
        out["conv%d.weight" % i] = torch.from_numpy(((2.0 * u - 1.0) * bound).astype(np.float32).reshape(cout, cin, 6, 6))
        if bias:
            ub = hash_uniform(cout, _key("patchgan.conv%d.bias" % i, seed))
            out["conv%d.bias" % i] = torch.from_numpy(((2.0 * ub - 1.0) * bound).astype(np.float32))
        if i in (2, 3, 4):
            n = "norm%d" % (i - 1)
            uw = hash_uniform(cout, _key("patchgan.%s.weight" % n, seed))
            ub = hash_uniform(cout, _key("patchgan.%s.bias" % n, seed))
            out[n + ".weight"] = torch.from_numpy((0.5 + uw).astype(np.float32))
            out[n + ".bias"] = torch.from_numpy((0.2 * (ub - 0.5)).astype(np.float32))
            out[n + ".running_mean"] = torch.zeros(cout)
            out[n + ".running_var"] = torch.ones(cout)
            out[n + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    # state_dict order of the module: conv1, conv2, norm1, conv3, norm2, conv4, norm3, conv5
    order = PatchGAN(input_nc, ndf).state_dict().keys()
    return OrderedDict((k, out[k]) for k in order)


def synth_gan_case(seed: int = 0, batch: int = 2, side: int = 224) -> Dict[str, torch.Tensor]:
    """Inputs of the PatchGAN fixture / tests from the counter-based hash (nothing stored): real and generated images,
    the other prediction images, ground truth, disk-shaped face / eye masks, latent-code offsets."""
    from .synth import _key, hash_uniform
    img = lambda name: torch.from_numpy(hash_uniform(batch * 3 * side * side, _key("gan." + name, seed)).astype(np.float32)
                                        .reshape(batch, 3, side, side))
    out = {k: img(k) for k in ("real_img", "fake_img", "merge_img_face", "merge_img_eyes", "bg_img", "gt")}
    yy, xx = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
    s = side / 224.0
    disk = lambda cy, cx, r: (((yy - cy * s) ** 2 + (xx - cx * s) ** 2) < (r * s) ** 2).float().view(1, 1, side, side).expand(batch, 1, side, side).clone()
    out["face"], out["leye"], out["reye"] = disk(112, 112, 84), disk(90, 78, 16), disk(90, 146, 16)
    out["full_eye"] = torch.clamp(out["leye"] + out["reye"], 0, 1)
    for name, n in (("iden", 100), ("expr", 79), ("appea", 127)):
        out["code_" + name] = torch.from_numpy((0.2 * (hash_uniform(batch * n, _key("gan.code." + name, seed)) - 0.5)).astype(np.float32).reshape(batch, n))
    return out


def min_abs_preactivation(d: PatchGAN, state, case) -> float:
    """Smallest |input| of the discriminator's four LeakyReLUs over the real and the generated batch, in fp64: the margin
    the fixture generator requires (a value within fp32 rounding of zero makes every gradient upstream of it depend on the
    summation order)."""
    d = PatchGAN(d.conv1.in_channels, d.conv1.out_channels)
    d.load_state_dict(state)
    d = d.double().train()
    lo = [float("inf")]
    hook = lambda m, inp: lo.__setitem__(0, min(lo[0], float(inp[0].detach().abs().min())))
    h = d.act.register_forward_pre_hook(hook)
    with torch.no_grad():
        d(case["real_img"].double()); d(case["fake_img"].double())
    h.remove()
    return lo[0]
