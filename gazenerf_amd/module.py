"""nn.Module surface of the hot path with the reference's parameter names.

``HotPathRenderer`` owns ``fg_CD_predictor_face`` / ``fg_CD_predictor_eyes`` (and optionally
``fine_fg_CD_predictor``) whose parameters are named and shaped exactly like the reference's
``MLPforNeRF`` (models/mlp_nerf.py:29-93: ``FeaExt_module_{0..7}``, ``density_module``,
``RGB_layer_{0,1,2}``, Conv2d weights ``[out,in,1,1]``), so
``renderer.load_state_dict(check_dict["net"], strict=False)`` fills them from a reference
checkpoint (trainer/gazenerf_trainer.py:116,173).  The arithmetic is libgnr's; these modules
only hold parameters.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import render as R_
from . import synth


class MLPParams(nn.Module):
    """Parameter container mirroring MLPforNeRF (models/mlp_nerf.py:13-93), same init rules."""

    def __init__(self, vp_channels: int, vd_channels: int, n_layers: int = 8, h_channel: int = 384,
                 res_nfeat: int = 258):
        super().__init__()
        self.vp_channels, self.vd_channels = vp_channels, vd_channels
        self.n_layers, self.h_channel, self.res_nfeat = n_layers, h_channel, res_nfeat
        skips = [n_layers // 2]
        self.add_module("FeaExt_module_0", nn.Conv2d(vp_channels, h_channel, 1))
        for i in range(n_layers - 1):
            cin = h_channel + vp_channels if i in skips else h_channel
            m = nn.Conv2d(cin, h_channel, 1)
            nn.init.xavier_uniform_(m.weight.data)
            self.add_module("FeaExt_module_%d" % (i + 1), m)
        m = nn.Conv2d(h_channel, 1, 1)
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data[:] = 0.0
        self.add_module("density_module", m)
        m = nn.Conv2d(h_channel, h_channel, 1)
        nn.init.xavier_uniform_(m.weight.data)
        self.add_module("RGB_layer_0", m)
        self.add_module("RGB_layer_1", nn.Conv2d(h_channel + vd_channels, h_channel // 2, 1))
        self.add_module("RGB_layer_2", nn.Conv2d(h_channel // 2, res_nfeat, 1))

    def param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in R_.PARAM_ORDER]

    def forward(self, *a, **k):
        raise RuntimeError("MLPParams only holds parameters; evaluate it through HotPathRenderer")


VD_FREQS = 4                       # GazeNeRFNet.vd_n_freqs (models/gaze_nerf.py:30)
VD_DIMS = 3 + 6 * VD_FREQS         # 27 channels with include_input (gaze_nerf.py:31, 72-76)


def view_direction_embedding(batch_xy, batch_Rmats, batch_inv_inmats):
    """[B, 27, N_r]: the reference's ``vd_encoder(fg_dirs)`` (models/gaze_nerf.py:240-241) for one sample of each
    ray -- ``dirs`` is the normalised ray direction expanded along the samples (utils/model_utils.py:366-369, 317),
    so the embedding is constant along a ray.  Embedder order: x | sin(2^k x), cos(2^k x) for k = 0..3
    (utils/model_utils.py:253-280).  Plain torch: differentiable w.r.t. the rotation."""
    xyz = torch.nn.functional.pad(batch_xy, [0, 0, 0, 1, 0, 0], mode="constant", value=1.0)
    d = batch_Rmats.bmm(batch_inv_inmats.bmm(xyz))
    d = d / torch.norm(d, dim=1, keepdim=True)
    parts = [d]
    for k in range(VD_FREQS):
        parts += [torch.sin(d * (2.0 ** k)), torch.cos(d * (2.0 ** k))]
    return torch.cat(parts, dim=1)


def view_direction_ray_bias(vd_embed, mlp: "MLPParams"):
    """[B, N_r, H/2] = W[:, H:H+27] @ embedding: the view-direction columns of RGB_layer_1 folded into a per-ray bias
    (render_two_stream(ray_bias_*=...)); autograd carries the gradient back into those columns and the rotation."""
    h = mlp.h_channel
    w = mlp.RGB_layer_1.weight[:, h:h + VD_DIMS, 0, 0]
    return torch.einsum("ok,bkr->bro", w, vd_embed).contiguous()


class HotPathRenderer(nn.Module):
    """GazeNeRF's two-stream volumetric renderer (the hot span of GazeNeRFNet._forward).

    forward(batch_xy, batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, appea_code, gaze_code,
            for_train=False, t_rand=None) -> dict(feat_face, bg_alpha_face, feat_eyes, bg_alpha_eyes, ...)

    ``for_train=True`` applies the stratified jitter of utils/model_utils.py:302-307; pass ``t_rand``
    to pin the draw, otherwise one is drawn with ``torch.rand`` on the device.
    With ``hier_sampling=True`` the intended fine pass runs as well:
    FineSample on the face weights -> third MLP over 64+128 samples -> "feat_fine"/"bg_alpha_fine".
    """

    def __init__(self, num_sample_coarse: int = 64, num_sample_fine: int = 128, world_z1: float = 2.5,
                 world_z2: float = -3.5, hidden: int = 384, featmap_nc: int = 258,
                 shape_dims: int = synth.SHAPE_DIMS, gaze_dims: int = synth.GAZE_DIMS,
                 appea_dims: int = synth.APPEA_DIMS, hier_sampling: bool = False, precision: str = "fp32",
                 ws_budget_bytes: Optional[int] = None, include_vd: bool = False, vd_fold: str = "device"):
        super().__init__()
        if vd_fold not in ("device", "torch"):
            raise ValueError("vd_fold must be 'device' or 'torch'")
        # include_vd: who folds the 27 view-direction columns of RGB_layer_1 into the per-ray bias the kernels take --
        # libgnr itself (round 3: gnr_vd.hip, forward and backward inside gnr_fwd / gnr_bwd) or this module in torch
        # (rounds 1-2: view_direction_ray_bias + autograd; kept as the cross-check)
        self.vd_fold = vd_fold
        self.ws_budget_bytes = ws_budget_bytes      # None == render.DEFAULT_WS_BUDGET; see render_two_stream
        # inference calls keep their workspace and skip the weight re-layout while the parameters are unchanged
        self._wcache, self._wcache_fine = R_.PackedWeightCache(), R_.PackedWeightCache()
        if precision not in ("fp32", "bf16x3"):
            raise ValueError("precision must be 'fp32' or 'bf16x3'")
        self.precision = precision
        self.num_sample_coarse, self.num_sample_fine = num_sample_coarse, num_sample_fine
        self.world_z1, self.world_z2 = world_z1, world_z2
        self.hidden, self.featmap_nc = hidden, featmap_nc
        self.hier_sampling = hier_sampling
        # include_vd (GazeNeRFNet's constructor argument, models/gaze_nerf.py:14, 70-80): RGB_layer_1 of every MLP also
        # sees the 27-channel embedding of the view direction, in front of the appearance code
        self.include_vd = include_vd
        self.vd_dims = VD_DIMS if include_vd else 0
        vp = 63 + shape_dims + gaze_dims
        vd_ch = appea_dims + self.vd_dims
        self.fg_CD_predictor_eyes = MLPParams(vp, vd_ch, h_channel=hidden, res_nfeat=featmap_nc)
        self.fg_CD_predictor_face = MLPParams(vp, vd_ch, h_channel=hidden, res_nfeat=featmap_nc)
        if hier_sampling:
            self.fine_fg_CD_predictor = MLPParams(vp, vd_ch, h_channel=hidden, res_nfeat=featmap_nc)

    # The packed-weight caches key on tensor identity + version counter + data_ptr; an in-place write through ``.data``
    # is invisible to all three (render.PackedWeightCache).  The usual places where parameters change under an
    # eval-mode module -- mode switches, device / dtype moves, checkpoint loads -- drop the caches outright.
    def clear_weight_caches(self):
        self._wcache.clear()
        self._wcache_fine.clear()

    def train(self, mode: bool = True):
        self.clear_weight_caches()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self.clear_weight_caches()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.clear_weight_caches()
        return super().load_state_dict(*args, **kwargs)

    def forward(self, batch_xy, batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, appea_code,
                gaze_code, for_train: bool = False, t_rand: Optional[torch.Tensor] = None,
                u_fine: Optional[torch.Tensor] = None, return_weights: bool = False):
        B, _, n_r = batch_xy.shape
        n_p = self.num_sample_coarse
        if for_train and t_rand is None:
            t_rand = torch.rand(B, n_r, n_p + 1, device=batch_xy.device)
        want_w = return_weights or self.hier_sampling
        rb_face = rb_eyes = rb_fine = None
        if self.include_vd and self.vd_fold == "torch":
            vd = view_direction_embedding(batch_xy, batch_Rmats, batch_inv_inmats)
            rb_face = view_direction_ray_bias(vd, self.fg_CD_predictor_face)
            rb_eyes = view_direction_ray_bias(vd, self.fg_CD_predictor_eyes)
            if self.hier_sampling:
                rb_fine = view_direction_ray_bias(vd, self.fine_fg_CD_predictor)
        out = R_.render_two_stream(
            batch_xy, batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, gaze_code, appea_code,
            self.fg_CD_predictor_face.param_list(), self.fg_CD_predictor_eyes.param_list(),
            n_samples=n_p, world_z1=self.world_z1, world_z2=self.world_z2, t_rand=t_rand,
            return_weights=want_w, hidden=self.hidden, feat_nc=self.featmap_nc, precision=self.precision,
            ws_budget_bytes=self.ws_budget_bytes, weight_cache=self._wcache,
            vd_dims=self.vd_dims, ray_bias_face=rb_face, ray_bias_eyes=rb_eyes)
        if self.hier_sampling:
            zv = R_.sample_zvals(batch_xy, batch_Rmats.detach(), batch_Tvecs.detach(), batch_inv_inmats,
                                 n_samples=n_p, world_z1=self.world_z1, world_z2=self.world_z2, t_rand=t_rand)
            if for_train and u_fine is None:
                u_fine = torch.rand(B * n_r, self.num_sample_fine + 1, device=batch_xy.device)
            edges = R_.importance_resample(out["w_face"], zv, n_fine=self.num_sample_fine, u=u_fine)
            fine = R_.render_two_stream(
                batch_xy, batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, gaze_code, appea_code,
                self.fine_fg_CD_predictor.param_list(), None,
                n_samples=n_p + self.num_sample_fine, world_z1=self.world_z1, world_z2=self.world_z2,
                z_edges=edges, edges_follow_T=True,       # FineSample detaches only the weights (model_utils.py:418)
                hidden=self.hidden, feat_nc=self.featmap_nc, precision=self.precision,
                ws_budget_bytes=self.ws_budget_bytes, weight_cache=self._wcache_fine,
                vd_dims=self.vd_dims, ray_bias_face=rb_fine)
            out["feat_fine"], out["bg_alpha_fine"] = fine["feat_face"], fine["bg_alpha_face"]
            out["fine_edges"] = edges
        return out


class GazeNeRFNetAMD(HotPathRenderer):
    """The reference's ``GazeNeRFNet`` (models/gaze_nerf.py) end to end on MI355X: hot path (render op) ->
    feature-map merge (gaze_nerf.py:164-203) -> ``NeuralRenderer`` x4 (:176,200,201,205).

    Same submodule / parameter names as the reference (``fg_CD_predictor_{face,eyes}.*``,
    ``neural_render.*`` including ``bg_featmap``), so ``load_state_dict(check_dict["net"])`` works strictly
    for the default (non-hierarchical) configuration.  ``forward`` takes the reference's arguments
    (gaze_nerf.py:318-331) plus ``t_rand`` to pin the train-mode jitter, and returns
    ``{"coarse_dict": {"merge_img_face", "merge_img_eyes", "merge_img", "bg_img"}}`` like the reference."""

    def __init__(self, featmap_size: int = 64, pred_img_size: int = 512, bg_type: str = "white", **kw):
        super().__init__(**kw)
        from .upsample import NeuralRendererAMD
        self.featmap_size, self.pred_img_size = featmap_size, pred_img_size
        self.neural_render = NeuralRendererAMD(bg_type=bg_type, feat_nc=self.featmap_nc, out_dim=3, final_actvn=True,
                                               min_feat=32, featmap_size=featmap_size, img_size=pred_img_size)

    def forward(self, mode, batch_xy, batch_uv, bg_code, shape_code, appea_code, gaze_code, batch_Rmats, batch_Tvecs,
                batch_inv_inmats, dist_expr=False, t_rand: Optional[torch.Tensor] = None, **kwargs):
        from .merge import merge_featmaps
        assert mode in ("train", "test") and bg_code is None
        out = HotPathRenderer.forward(self, batch_xy, batch_Rmats, batch_Tvecs, batch_inv_inmats, shape_code, appea_code,
                                      gaze_code, for_train=(mode == "train"), t_rand=t_rand)
        B, S, C = batch_xy.shape[0], self.featmap_size, self.featmap_nc
        if batch_xy.shape[2] != S * S:
            raise ValueError("batch_xy must cover the %dx%d feature map" % (S, S))
        bg = self.neural_render.get_bg_featmap()
        mf, ep, m = merge_featmaps(out["feat_face"], out["bg_alpha_face"], out["feat_eyes"], out["bg_alpha_eyes"],
                                   bg.reshape(1, C, S * S), gaze_code.reshape(-1, 2))
        # the reference calls NeuralRenderer four times with the same weights (gaze_nerf.py:176,200,201,205): one
        # call on the concatenated batch fills the chip better and quarters the launches; results are per image
        stacked = torch.cat([mf.reshape(B, C, S, S), ep.reshape(B, C, S, S), m.reshape(B, C, S, S), bg.reshape(1, C, S, S)], dim=0)
        imgs = self.neural_render(stacked)
        res = {"merge_img_face": imgs[:B], "merge_img_eyes": imgs[B:2 * B], "merge_img": imgs[2 * B:3 * B], "bg_img": imgs[3 * B:]}
        return {"coarse_dict": res}
