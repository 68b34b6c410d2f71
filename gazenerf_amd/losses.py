"""Image-space losses and the per-sample code / camera assembly of the reference's training step
(the parts that need no pretrained network).

Restated from losses/gazenerf_loss.py:294-470 (``calc_data_loss`` / ``calc_total_loss`` with
every switch of the reference: the PatchGAN term needs no pretrained network -- ``gazenerf_amd.gan`` -- and is added when a
discriminator is passed; the VGG-perceptual terms are added when a ``gazenerf_amd.perceptual.VGGPerceptualLoss`` is passed
-- its ImageNet weights are the caller's to supply; the gaze-angular term when a ``perceptual.GazeAngularLoss`` is passed --
the gaze estimator's trained weights are the caller's too) and trainer/base.py:92-124,
trainer/gazenerf_trainer.py:338-405 (``eulurangle2Rmat``, ``build_code_and_cam``).  Plain PyTorch on the GPU:
these are a few reductions over [B,3,512,512] images, not a hot path.  Masked means are computed as
sum(mask * err) / count instead of boolean indexing (same value up to summation order, no host sync).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def euler_to_rotation(angles: torch.Tensor) -> torch.Tensor:
    """[B,3] (x, y, z) -> [B,3,3] = Rz @ Ry @ Rx  (trainer/base.py:92-124)."""
    s, c = torch.sin(angles), torch.cos(angles)
    one, zero = torch.ones_like(s[:, 0]), torch.zeros_like(s[:, 0])
    rx = torch.stack([one, zero, zero, zero, c[:, 0], -s[:, 0], zero, s[:, 0], c[:, 0]], dim=1).view(-1, 3, 3)
    ry = torch.stack([c[:, 1], zero, s[:, 1], zero, one, zero, -s[:, 1], zero, c[:, 1]], dim=1).view(-1, 3, 3)
    rz = torch.stack([c[:, 2], -s[:, 2], zero, s[:, 2], c[:, 2], zero, zero, zero, one], dim=1).view(-1, 3, 3)
    return rz.bmm(ry.bmm(rx))


def region_masks(face_mask, full_eye_mask, left_eye_mask, right_eye_mask):
    """gazenerf_loss.py:438-443.  Inputs [B,1,H,W] float masks; returns boolean head / face / eyes / nonhead."""
    head = torch.logical_and(face_mask >= 0.5, full_eye_mask < 0.5)
    face = torch.logical_and(face_mask >= 0.5, torch.logical_and(left_eye_mask < 0.5, right_eye_mask < 0.5))
    eyes = torch.logical_or(left_eye_mask >= 0.5, right_eye_mask >= 0.5)
    return {"head": head, "face": face, "eyes": eyes, "nonhead": face_mask < 0.5}


def _masked_mean(err, mask_c1):
    m = mask_c1.expand(-1, err.shape[1], -1, -1).to(err.dtype)
    return (err * m).sum() / m.sum()


def data_losses(pred: Dict[str, torch.Tensor], gt_rgb, masks, bg_value: float = 1.0, use_l1: bool = False,
                epoch: int = 0, discriminator=None, batch_num: int = 0, vgg=None, vgg_importance: float = 1.0,
                gaze=None, eye_loss_importance: float = 1.0):
    """gazenerf_loss.py:294-403; with ``gaze`` (a ``perceptual.GazeAngularLoss``) the angular term of :383-391; with ``vgg`` (a ``perceptual.VGGPerceptualLoss``) the three
    perceptual terms of :360-381; with ``discriminator`` (a ``gan.PatchGAN`` whose parameters the caller has frozen, as the
    trainer does) the generator's PatchGAN term of :396-401.  The entries are in the reference's order (their sum is)."""
    pen = (lambda d: d.abs()) if use_l1 else (lambda d: d * d)
    res = {
        "bg_loss": torch.mean((pred["bg_img"] - bg_value) ** 2),
        "eyes_loss": _masked_mean(pen(pred["merge_img_eyes"] - gt_rgb), masks["eyes"]),
        "face_loss": _masked_mean(pen(pred["merge_img_face"] - gt_rgb), masks["face"]),
        "nonhead_loss": _masked_mean((pred["merge_img"] - bg_value) ** 2, masks["nonhead"]),
    }
    if epoch > -1:
        res["head_loss"] = _masked_mean(pen(pred["merge_img"] - gt_rgb), masks["head"])
    if vgg is not None:
        from .perceptual import vgg_terms
        res.update(vgg_terms(vgg, pred, gt_rgb, masks, bg_value, vgg_importance))
    if gaze is not None and epoch > -1:
        from .perceptual import angular_term
        res["angular"] = angular_term(gaze, pred, gt_rgb, masks, bg_value, eye_loss_importance)
    if discriminator is not None:
        from .gan import generator_term
        res["gen_patch_gan_loss"] = generator_term(discriminator, pred["merge_img"], epoch, batch_num)
    return res


def total_loss(pred, gt_rgb, face_mask, full_eye_mask, left_eye_mask, right_eye_mask, opt_codes,
               delta_cam: Optional[Dict[str, torch.Tensor]] = None, bg_value: float = 1.0, use_l1: bool = False,
               epoch: int = 0, discriminator=None, batch_num: int = 0, vgg=None, vgg_importance: float = 1.0,
               gaze=None, eye_loss_importance: float = 1.0):
    """gazenerf_loss.py:405-470: data terms + 0.001 |delta cam|^2 + code regularisers (0.001 iden, 1.0 expr,
    0.001 appea, 0.01 bg).  ``pred`` is the network's ``coarse_dict``."""
    masks = region_masks(face_mask, full_eye_mask, left_eye_mask, right_eye_mask)
    loss = data_losses(pred, gt_rgb, masks, bg_value, use_l1, epoch, discriminator, batch_num, vgg, vgg_importance, gaze,
                       eye_loss_importance)
    total = sum(loss.values())
    if delta_cam is not None:
        loss["delta_eular"] = torch.mean(delta_cam["delta_eulur"] ** 2)
        loss["delta_tvec"] = torch.mean(delta_cam["delta_tvec"] ** 2)
        total = total + 0.001 * loss["delta_eular"] + 0.001 * loss["delta_tvec"]
    loss["iden_code"] = torch.mean(opt_codes["iden"] ** 2)
    loss["expr_code"] = torch.mean(opt_codes["expr"] ** 2)
    loss["appea_code"] = torch.mean(opt_codes["appea"] ** 2)
    bg = opt_codes.get("bg")
    loss["bg_code"] = torch.mean(bg ** 2) if bg is not None else torch.zeros_like(loss["iden_code"])
    total = total + 0.001 * loss["iden_code"] + 1.0 * loss["expr_code"] + 0.001 * loss["appea_code"] + 0.01 * loss["bg_code"]
    loss["total_loss"] = total
    return loss


class Fitter:
    """The reference's per-sample fitting step (gazenerf_trainer.py:338-405, 425-476, 478-534) around
    ``GazeNeRFNetAMD``: learnable identity / expression / appearance offsets and camera deltas per dataset row,
    Adam with the reference's learning-rate ratios (net 1, iden 1, expr 0.1, appea 1, camera deltas 0.1).

    ``base`` holds what the dataset provides per batch row: iden [B,100], expr [B,79], text [B,100], illu [B,27]
    (shape_code = [iden|expr], appea_code = [text|illu]), gaze [B,2], c2w_Rmat [B,3,3], c2w_Tvec [B,3,1],
    inv_inmat [B,3,3]."""

    def __init__(self, net, n_rows: int, lr: float = 1e-4, opt_cam: bool = True, device=None):
        dev = device or next(net.parameters()).device
        self.net, self.opt_cam = net, opt_cam
        z = lambda *s: torch.zeros(*s, device=dev, requires_grad=True)
        self.iden_offset, self.expr_offset, self.appea_offset = z(n_rows, 100), z(n_rows, 79), z(n_rows, 127)
        self.delta_EulurAngles, self.delta_Tvecs = z(n_rows, 3), z(n_rows, 3, 1)
        groups = [{"params": net.parameters(), "lr": lr}, {"params": [self.iden_offset], "lr": lr},
                  {"params": [self.expr_offset], "lr": lr * 0.1}, {"params": [self.appea_offset], "lr": lr}]
        if opt_cam:
            groups += [{"params": [self.delta_EulurAngles], "lr": lr * 0.1}, {"params": [self.delta_Tvecs], "lr": lr * 0.1}]
        self.optimizer = torch.optim.Adam(groups, betas=(0.9, 0.999))

    def build_code_and_cam(self, rows: slice, base):
        shape_code = torch.cat([base["iden"] + self.iden_offset[rows], base["expr"] + self.expr_offset[rows]], dim=-1)
        appea_code = torch.cat([base["text"], base["illu"]], dim=-1) + self.appea_offset[rows]
        opt_codes = {"bg": None, "iden": self.iden_offset[rows], "expr": self.expr_offset[rows], "appea": self.appea_offset[rows]}
        if self.opt_cam:
            dR = euler_to_rotation(self.delta_EulurAngles[rows])
            R = dR.bmm(base["c2w_Rmat"])
            T = dR.bmm(base["c2w_Tvec"]) + self.delta_Tvecs[rows]
            delta = {"delta_eulur": self.delta_EulurAngles[rows], "delta_tvec": self.delta_Tvecs[rows]}
        else:
            R, T, delta = base["c2w_Rmat"], base["c2w_Tvec"], None
        return shape_code, appea_code, base["gaze"], R, T, opt_codes, delta

    def step(self, rows: slice, xy, base, gt_rgb, face_mask, full_eye_mask, left_eye_mask, right_eye_mask,
             t_rand=None, epoch: int = 0, gan=None, batch_num: int = 0, vgg=None, vgg_importance: float = 1.0,
             gaze_loss=None, eye_loss_importance: float = 1.0):
        """``vgg`` / ``gaze_loss``: ``perceptual.VGGPerceptualLoss`` / ``perceptual.GazeAngularLoss`` modules (``use_vgg_loss`` /
        ``use_angular_loss``).  ``gan``: a ``gazenerf_amd.gan.DiscriminatorStep`` (``use_patch_gan_loss``): the discriminator is updated on
        (ground truth, detached prediction) first, then frozen while the generator term joins the total loss
        (gazenerf_trainer.py:487-528)."""
        shape_code, appea_code, gaze, R, T, opt_codes, delta = self.build_code_and_cam(rows, base)
        pred = self.net("train", xy, None, None, shape_code, appea_code, gaze, R, T, base["inv_inmat"], t_rand=t_rand)
        extra = {}
        if gan is not None:
            extra = gan.step(gt_rgb, face_mask, pred["coarse_dict"]["merge_img"])
        losses = total_loss(pred["coarse_dict"], gt_rgb, face_mask, full_eye_mask, left_eye_mask, right_eye_mask,
                            opt_codes, delta, epoch=epoch, discriminator=gan.discriminator if gan is not None else None,
                            batch_num=batch_num, vgg=vgg, vgg_importance=vgg_importance, gaze=gaze_loss,
                            eye_loss_importance=eye_loss_importance)
        self.optimizer.zero_grad()
        losses["total_loss"].backward()
        self.optimizer.step()
        out = {k: float(v.detach()) for k, v in losses.items()}
        out.update({k: float(v) for k, v in extra.items()})
        return out
