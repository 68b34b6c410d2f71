"""ORACLE -- test infrastructure, NOT product code.

A plain PyTorch CPU restatement of the GazeNeRF volumetric-render hot path
(SURVEY.md section 8(a), rows A1-A6).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this file; nothing under
``gazenerf_amd/`` does.  The product path is the HIP library; it has no CPU fallback.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference's own modules
in the build container, asserts every function below equals them (<=1e-6 max-abs,
observed 0 for most stages) and writes the fixtures under ``tests/golden/`` that
``tests/test_oracle_golden.py`` replays wherever the reference is absent.

Each function cites the reference lines it restates (paths relative to the
reference repository root).  Tensors keep the reference's channels-first layout
``[B, C, N_r, N_p]`` so that outputs compare without reshuffling.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- A1
def sample_edges(xy, R, T, Kinv, n_samples, world_z1=2.5, world_z2=-3.5, t_rand=None):
    """Ray geometry + the N_p+1 sample edges (before the edge->sample rule), model_utils.py:364-372,
    339-357, 302-307.  Returns (edges [B,N_r,N_p+1], ray_o, ray_d, ray_l as [B,C,N_r,1])."""
    xyz = F.pad(xy, [0, 0, 0, 1, 0, 0], mode="constant", value=1.0)          # :365
    ray_d = R.bmm(Kinv.bmm(xyz))                                              # :366
    ray_l = torch.norm(ray_d, dim=1, keepdim=True)                            # :367
    ray_d = ray_d / ray_l                                                     # :368
    ray_l = -1.0 / ray_d[:, -1:, :]                                           # :369
    B, _, n_r = xy.shape
    ray_o = T.expand(B, 3, n_r)                                               # :372

    rela_z1 = (ray_o[:, -1, :] - world_z1).unsqueeze(-1)                      # :339-343
    rela_z2 = (ray_o[:, -1, :] - world_z2).unsqueeze(-1)
    ray_o = ray_o.unsqueeze(-1)
    ray_d = ray_d.unsqueeze(-1)
    ray_l = ray_l.unsqueeze(-1)
    t = torch.linspace(0.0, 1.0, steps=n_samples + 1, dtype=xy.dtype).view(1, 1, -1)  # :352-354
    zvals = rela_z1 * (1.0 - t) + rela_z2 * t                                 # :355-357

    if t_rand is not None:                                                    # :302-307
        mids = 0.5 * (zvals[:, :, 1:] + zvals[:, :, :-1])
        upper = torch.cat([mids, zvals[:, :, -1:]], dim=-1)
        lower = torch.cat([zvals[:, :, :1], mids], dim=-1)
        zvals = lower + (upper - lower) * t_rand
    return zvals, ray_o, ray_d, ray_l


def gen_sample_points(xy, R, T, Kinv, n_samples, world_z1=2.5, world_z2=-3.5, t_rand=None,
                      z_edges=None):
    """utils/model_utils.py:364-375 (forward), 332-362, 291-330.

    xy [B,2,N_r]; R [B,3,3] cam-to-world; T [B,3,1]; Kinv [B,3,3].
    ``t_rand`` [B,N_r,n_samples+1] replaces ``torch.rand_like`` (model_utils.py:306);
    ``None`` is the reference's ``disturb=False``.  ``z_edges`` overrides the edges (the fine
    pass, model_utils.py:476-488).
    """
    zvals, ray_o, ray_d, ray_l = sample_edges(xy, R, T, Kinv, n_samples, world_z1, world_z2, t_rand)
    if z_edges is not None:
        zvals = z_edges
    return points_from_zvals(zvals, ray_o, ray_d, ray_l)


def points_from_zvals(zvals, ray_o, ray_d, ray_l):
    """utils/model_utils.py:309-328 (shared tail of GenSamplePoints and FineSample:394-409)."""
    z_dists = zvals[:, :, 1:] - zvals[:, :, :-1]
    z_dists = z_dists.unsqueeze(1) * ray_l                                    # :310
    zv = zvals[:, :, :-1].unsqueeze(1)                                        # :312-313
    pts = ray_o + ray_d * ray_l * zv                                          # :315
    return {
        "pts": pts, "dirs": ray_d.expand(-1, -1, -1, zv.size(-1)),
        "zvals": zv, "z_dists": z_dists,
        "batch_ray_o": ray_o, "batch_ray_d": ray_d, "batch_ray_l": ray_l,
    }


# --------------------------------------------------------------------------- A2
def embed(x, n_freqs=10, include_input=True):
    """utils/model_utils.py:253-280.  [B,3,N_r,N_p] -> [B,3+6*n_freqs,N_r,N_p];
    channel order x | sin(2^0 x) cos(2^0 x) | sin(2^1 x) cos(2^1 x) | ..."""
    freqs = 2.0 ** torch.linspace(0.0, n_freqs - 1, steps=n_freqs)
    res = [x] if include_input else []
    for f in freqs:
        res.append(torch.sin(x * f))
        res.append(torch.cos(x * f))
    return torch.cat(res, dim=1)


# --------------------------------------------------------------------------- A4
def _fc(params, name, x):
    w = params[name + ".weight"]
    return F.conv2d(x, w.to(x.dtype), params[name + ".bias"].to(x.dtype))


def mlp_forward(params, vp_in, vd_in, n_layers=8):
    """models/mlp_nerf.py:95-119.  Returns (feat [B,C_f,N_r,N_p], sigma [B,1,N_r,N_p]).
    ``res_nfeat != 3`` so no sigmoid (mlp_nerf.py:116-117)."""
    x = vp_in
    for i in range(n_layers):
        x = F.relu(_fc(params, "FeaExt_module_%d" % i, x))
        if i == n_layers // 2:
            x = torch.cat([vp_in, x], dim=1)                                  # :107
    density = _fc(params, "density_module", x)
    x = _fc(params, "RGB_layer_0", x)                                         # no activation, :110
    x = F.relu(_fc(params, "RGB_layer_1", torch.cat([x, vd_in], dim=1)))
    feat = _fc(params, "RGB_layer_2", x)
    if params["RGB_layer_2.weight"].shape[0] == 3:
        feat = torch.sigmoid(feat)
    return feat, F.relu(density)


# --------------------------------------------------------------------------- A5
def calc_ray_color(feat, sigma, dists, zvals):
    """utils/model_utils.py:498-534.  -> feat_out [B,C,N_r], bg_alpha [B,1,N_r],
    depth [B,1,N_r], weights [B,1,N_r,N_p]."""
    alpha = 1.0 - torch.exp(-sigma * dists)                                   # :500
    x = 1.0 - alpha + 1e-10                                                   # :508
    x = F.pad(x, [1, 0, 0, 0, 0, 0, 0, 0], mode="constant", value=1.0)
    x = torch.cumprod(x, dim=-1)
    w = alpha * x[:, :, :, :-1]                                               # :512
    feat_out = torch.sum(w * feat, dim=-1)
    depth = torch.sum(w * zvals, dim=-1)
    bg_alpha = 1.0 - torch.sum(w, dim=-1)
    return feat_out, bg_alpha, depth, w


# --------------------------------------------------------------------------- A6
def fine_sample(weights, sample_dict, n_fine, u=None):
    """utils/model_utils.py:413-490.  ``n_fine`` is ``opt.num_sample_fine`` (128); the
    reference draws n_fine+1 values (:381).  ``u`` [N_t, n_fine+1] replaces torch.rand."""
    nf = n_fine + 1
    coarse_z = sample_dict["zvals"]
    tw = weights[:, :, :, 1:-1].detach()
    B, _, n_r, nc2 = tw.shape
    tw = tw.reshape(-1, nc2)
    pdf = tw / torch.sum(tw + 1e-5, dim=-1, keepdim=True)
    cdf = F.pad(torch.cumsum(pdf, dim=-1), [1, 0, 0, 0], value=0.0).contiguous()
    n_t = cdf.size(0)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=nf, dtype=weights.dtype).view(1, nf).expand(n_t, nf)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=nc2)
    cz = coarse_z.reshape(n_t, nc2 + 2)
    bins = 0.5 * (cz[:, 1:] + cz[:, :-1])
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    fz = bin_b + t * (bin_a - bin_b)
    fz, _ = torch.sort(torch.cat([cz, fz], dim=-1), dim=-1)
    fz = fz.view(B, n_r, nf + nc2 + 2)
    res = points_from_zvals(fz, sample_dict["batch_ray_o"], sample_dict["batch_ray_d"],
                            sample_dict["batch_ray_l"])
    return {k: res[k] for k in ("pts", "dirs", "zvals", "z_dists")}


# ------------------------------------------------------------------ A1..A5 chained
def _stream(params, pts_embed, shape_ext, appea, dists, zvals, vd_embed=None):
    """models/gaze_nerf.py:136-162 for one stream.  ``vd_embed`` [B,27,N_r,N_p]: the view-direction embedding of the
    ``include_vd`` option, concatenated IN FRONT of the appearance code (gaze_nerf.py:140-143)."""
    B, _, n_r, n_p = pts_embed.shape
    vp_in = torch.cat([pts_embed, shape_ext.view(B, -1, 1, 1).expand(-1, -1, n_r, n_p)], dim=1)
    vd_in = appea.view(B, -1, 1, 1).expand(-1, -1, n_r, n_p)
    if vd_embed is not None:
        vd_in = torch.cat([vd_embed, vd_in], dim=1)
    feat, sigma = mlp_forward(params, vp_in, vd_in)
    return calc_ray_color(feat, sigma, dists, zvals)


def render_two_stream(xy, R, T, Kinv, shape_code, gaze, appea_code, face_params, eyes_params,
                      n_samples, world_z1=2.5, world_z2=-3.5, t_rand=None, z_edges=None, include_vd=False):
    """models/gaze_nerf.py:231-262 + 136-162: the whole hot path for both streams.

    Returns a dict: feat_face/feat_eyes [B,C_f,N_r], bg_alpha_* [B,1,N_r], depth_* [B,1,N_r],
    w_face/w_eyes [B,1,N_r,N_p] and the sample dict under "samples".
    ``include_vd``: the sample directions go through the 4-frequency Embedder (gaze_nerf.py:30-31, 71-80, 240-243:
    27 channels) into RGB_layer_1 of both MLPs.
    """
    sd = gen_sample_points(xy, R, T, Kinv, n_samples, world_z1, world_z2, t_rand, z_edges)
    emb = embed(sd["pts"])
    vd = embed(sd["dirs"], n_freqs=4) if include_vd else None                 # gaze_nerf.py:240-241
    shape_ext = torch.cat([shape_code, gaze], dim=1)                          # gaze_nerf.py:248
    out = {"samples": sd}
    for tag, params in (("face", face_params), ("eyes", eyes_params)):
        f, a, d, w = _stream(params, emb, shape_ext, appea_code, sd["z_dists"], sd["zvals"], vd)
        out["feat_" + tag], out["bg_alpha_" + tag] = f, a
        out["depth_" + tag], out["w_" + tag] = d, w
    return out


def hier_fine_pass(coarse, shape_code, gaze, appea_code, fine_params, n_fine, u=None):
    """The *intended* fine pass of the dead branch models/gaze_nerf.py:282-318 composed from
    working sub-modules: FineSample on the face weights -> embed ->
    third MLP -> CalcRayColor."""
    fs = fine_sample(coarse["w_face"], coarse["samples"], n_fine, u)
    emb = embed(fs["pts"])
    shape_ext = torch.cat([shape_code, gaze], dim=1)
    f, a, d, w = _stream(fine_params, emb, shape_ext, appea_code, fs["z_dists"], fs["zvals"])
    return {"feat_fine": f, "bg_alpha_fine": a, "depth_fine": d, "w_fine": w, "samples": fs}


# --------------------------------------------------------------------------- N2
def rotation_matrix_2d(label):
    """utils/model_utils.py:11-29: [B,2] (pitch, yaw) -> [B,3,3] = M2(yaw) @ M1(pitch)."""
    cos, sin = torch.cos(label), torch.sin(label)
    ones, zeros = torch.ones_like(cos[:, 0]), torch.zeros_like(cos[:, 0])
    m1 = torch.stack([ones, zeros, zeros, zeros, cos[:, 0], -sin[:, 0], zeros, sin[:, 0], cos[:, 0]], dim=1).view(-1, 3, 3)
    m2 = torch.stack([cos[:, 1], zeros, sin[:, 1], zeros, ones, zeros, -sin[:, 1], zeros, cos[:, 1]], dim=1).view(-1, 3, 3)
    return torch.matmul(m2, m1)


def merge_featmaps(feat_face, bg_alpha_face, feat_eyes, bg_alpha_eyes, bg_featmap, gaze):
    """models/gaze_nerf.py:175-203 with utils/model_utils.py:32-46 (rotate): background blend, rotation
    of the eye feature triplets by the gaze, elementwise max.  Maps are [B,C,H,W] (C = 3*86),
    bg_alpha [B,1,H,W], bg_featmap [1,C,H,W], gaze [B,2].  Returns (merge_face, eyes_planes, merge)."""
    merge_face = feat_face + bg_alpha_face * bg_featmap                                 # :178
    merge_eyes = feat_eyes + bg_alpha_eyes * bg_featmap                                 # :179
    B, C, H, W = merge_eyes.shape
    emb = merge_eyes.clone().reshape(B, C // 3, 3, H, W)                                # :181-190
    rot = rotation_matrix_2d(gaze.reshape(-1, 2))                                       # model_utils.py:37-39
    x = torch.transpose(torch.transpose(emb, 2, 3), 3, 4)                               # [B,G,H,W,3]
    x = torch.matmul(x, rot.view(B, 1, 1, 3, 3))                                        # model_utils.py:42-44
    eyes_planes = torch.transpose(torch.transpose(x, 4, 3), 3, 2).reshape(B, C, H, W)   # :181-197
    return merge_face, eyes_planes, torch.maximum(merge_face, eyes_planes)             # :203


# --------------------------------------------------------------------------- N1
def kornia_filter2d(x, kernel, normalized=True):
    """kornia.filters.filter2d of kornia 0.6.4 (requirements.txt:9; absent here, so restated from its
    published source kornia/filters/filter.py -- Blur parity is UNPINNED by the reference):
    kernel [1,kh,kw]; normalize_kernel2d divides by sum|k|; input padded with border_type='reflect' by
    (kh//2, kw//2); depthwise cross-correlation."""
    k = kernel.to(x)
    if normalized:
        k = k / k.abs().sum(dim=(-2, -1), keepdim=True)
    c = x.shape[1]
    kh, kw = k.shape[-2:]
    xp = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="reflect")
    return F.conv2d(xp, k.reshape(1, 1, kh, kw).expand(c, 1, kh, kw), groups=c)


def blur(x):
    """models/pixel_shuffle_upsample.py:7-16: f = [1,2,1]; filter2d(x, f (x) f, normalized=True)."""
    f = torch.tensor([1.0, 2.0, 1.0], dtype=x.dtype)
    return kornia_filter2d(x, f[None, None, :] * f[None, :, None], normalized=True)


def _conv(params, name, x):
    return F.conv2d(x, params[name + ".weight"].to(x.dtype), params[name + ".bias"].to(x.dtype))


def pixel_shuffle_upsample(params, prefix, x):
    """models/pixel_shuffle_upsample.py:33-42."""
    y = x.repeat(1, 4, 1, 1)
    out = F.leaky_relu(_conv(params, prefix + ".layer_1", x), 0.2)
    out = F.leaky_relu(_conv(params, prefix + ".layer_2", out), 0.2)
    return blur(F.pixel_shuffle(out + y, 2))


def rgb_upsample(x):
    """models/neural_renderer.py:65-67: bilinear x2 (align_corners=False) then Blur."""
    return blur(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False))


def neural_renderer(params, x, n_blocks=3, final_actvn=True):
    """NeuralRenderer.forward, models/neural_renderer.py:100-113."""
    rgb = rgb_upsample(_conv(params, "feat_2_rgb_list.0", x))
    net = x
    for i in range(n_blocks):
        hid = _conv(params, "feat_layers.%d" % i, pixel_shuffle_upsample(params, "feat_upsample_list.%d" % i, net))
        net = F.leaky_relu(hid, 0.2)
        rgb = rgb + _conv(params, "feat_2_rgb_list.%d" % (i + 1), net)
        if i < n_blocks - 1:
            rgb = rgb_upsample(rgb)
    return torch.sigmoid(rgb) if final_actvn else rgb


def synthetic_loss(out):
    """loss = sum_streams(mean(feat^2) + mean(bg_alpha))."""
    return sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes"))
