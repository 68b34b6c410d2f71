"""Synthetic input recipe for the GazeNeRF volumetric-render hot path.

Everything a caller needs to drive ``render_two_stream`` without a dataset or a
checkpoint: the pixel grid and scaled inverse intrinsics (reference
utils/render_utils.py:20-40), the frontal and orbit cameras
(utils/render_utils.py:42-99), latent codes in the range real codes occupy, and
reference-shaped MLP parameters (models/mlp_nerf.py:29-93) filled from a
counter-based integer hash so that any host reproduces the same numbers
without shipping 12 MB of weights.

All functions return CPU float32 torch tensors (numpy inside).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

# configs/gazenerf_options.py:1-35
SHAPE_DIMS = 179          # iden(100) + expr(79)
GAZE_DIMS = 2
APPEA_DIMS = 127          # text(100) + illu(27)
N_FREQS = 10
EMBED_CH = 3 + 6 * N_FREQS            # 63
VP_CH = EMBED_CH + SHAPE_DIMS + GAZE_DIMS   # 244
HIDDEN = 384
FEAT_NC = 258
WORLD_Z1 = 2.5
WORLD_Z2 = -3.5

# configs/config_files/cam_inmat_info_32x32.json ("inv_inmat")
_INV_FOCAL_32 = 0.007790804840624332
_INV_CX = -0.12553827464580536
_INV_CY = -0.12832458317279816


def pixel_grid(side: int) -> torch.Tensor:
    """ray_xy [1,2,side*side]: row 0 = x = i % W, row 1 = y = i // W (render_utils.py:24-32)."""
    idx = np.arange(side * side)
    xy = np.stack([idx % side, idx // side], axis=0).astype(np.float32)
    return torch.from_numpy(xy).unsqueeze(0)


def scaled_kinv(side: int) -> torch.Tensor:
    """inv_inmat [1,3,3] with the focal terms divided by side/32 (render_utils.py:36-40)."""
    k = np.array([[_INV_FOCAL_32, 0.0, _INV_CX],
                  [0.0, _INV_FOCAL_32, _INV_CY],
                  [0.0, 0.0, 1.0]], dtype=np.float32)
    k[:2, :2] /= np.float32(side / 32.0)
    return torch.from_numpy(k).view(1, 3, 3)


def frontal_camera():
    """base_cam_info: R = diag(1,-1,-1), T = (0,0,12) (render_utils.py:88-91)."""
    r = torch.eye(3, dtype=torch.float32).view(1, 3, 3).clone()
    r[0, 1:, :] *= -1
    t = torch.zeros(1, 3, 1, dtype=torch.float32)
    t[0, 2, 0] = 12.0
    return r, t


def orbit_camera(view: int, view_num: int = 45):
    """One of the synthetic orbit cameras (render_utils.py:42-86)."""
    tv_z, tv_x = 12.0, 5.3
    center = np.zeros(3)
    radius = math.sqrt(np.sum((np.array([tv_x, 0.0, tv_z]) - center) ** 2)
                       - np.sum((np.array([0.0, 0.0, tv_z]) - center) ** 2))
    angle = np.linspace(0, 360.0, view_num)[view]
    theta = angle / 180.0 * 3.1415926535
    vp = np.array([math.cos(theta) * radius, math.sin(theta) * radius, tv_z])
    d1 = center - vp
    d2 = np.cross(np.array([0.0, -1.0, 0.0]), d1)
    d3 = np.cross(d1, d2)
    d1, d2, d3 = (v / np.linalg.norm(v) for v in (d1, d2, d3))
    r = np.zeros((3, 3), dtype=np.float32)
    r[:, 0], r[:, 1], r[:, 2] = d2, d3, d1
    return (torch.from_numpy(r).view(1, 3, 3),
            torch.from_numpy(vp.astype(np.float32)).view(1, 3, 1))


# ----------------------------------------------------------------------------
# counter-based hash -> uniform floats (splitmix64 finaliser)
# ----------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x.copy()
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(n: int, key: int) -> np.ndarray:
    """n floats in [0,1), float64, a pure function of (key, index)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(key) << np.uint64(32))
        bits = _splitmix64(_splitmix64(ctr))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _key(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in (name + "#%d" % seed).encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFF
    return h


def mlp_param_shapes(hidden: int = HIDDEN, vp_ch: int = VP_CH, vd_ch: int = APPEA_DIMS,
                     feat_nc: int = FEAT_NC, n_layers: int = 8):
    """Parameter names/shapes of one MLPforNeRF (models/mlp_nerf.py:29-93), Conv2d 1x1 layout."""
    shapes = OrderedDict()
    shapes["FeaExt_module_0"] = (hidden, vp_ch)
    for i in range(n_layers - 1):
        cin = hidden + vp_ch if i == n_layers // 2 else hidden
        shapes["FeaExt_module_%d" % (i + 1)] = (hidden, cin)
    shapes["density_module"] = (1, hidden)
    shapes["RGB_layer_0"] = (hidden, hidden)
    shapes["RGB_layer_1"] = (hidden // 2, hidden + vd_ch)
    shapes["RGB_layer_2"] = (feat_nc, hidden // 2)
    return shapes


_DEFAULT_INIT = ("FeaExt_module_0", "RGB_layer_1", "RGB_layer_2")   # mlp_nerf.py:61,66,75


def hash_mlp_params(stream: str, seed: int = 0, hidden: int = HIDDEN, vp_ch: int = VP_CH,
                    vd_ch: int = APPEA_DIMS, feat_nc: int = FEAT_NC,
                    density_scale: float = 1.0, bias_scale: float = 1.0):
    """Reference-shaped parameters drawn from the reference's init *distributions*.

    Xavier-uniform for the layers mlp_nerf.py passes to _xavier_init, PyTorch's Conv2d
    default (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))) for the other three,
    biases U(+-1/sqrt(fan_in)) except density_module.bias = 0 (mlp_nerf.py:66).
    ``density_scale`` multiplies density_module.weight (the "opaque head" variant, x50).
    """
    out = OrderedDict()
    for name, (cout, cin) in mlp_param_shapes(hidden, vp_ch, vd_ch, feat_nc).items():
        if name in _DEFAULT_INIT:
            bound = 1.0 / math.sqrt(cin)
        else:
            bound = math.sqrt(6.0 / (cin + cout))
        u = hash_uniform(cout * cin, _key(stream + "." + name + ".weight", seed))
        w = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(cout, cin, 1, 1)
        if name == "density_module":
            w = w * np.float32(density_scale)
            b = np.zeros(cout, dtype=np.float32)
        else:
            ub = hash_uniform(cout, _key(stream + "." + name + ".bias", seed))
            b = ((2.0 * ub - 1.0) * (bias_scale / math.sqrt(cin))).astype(np.float32)
        out[name + ".weight"] = torch.from_numpy(w)
        out[name + ".bias"] = torch.from_numpy(b)
    return out


def renderer_param_shapes(feat_nc: int = FEAT_NC, n_blocks: int = 3, min_feat: int = 32, out_dim: int = 3):
    """Parameter names/shapes of NeuralRenderer (models/neural_renderer.py:57-98) minus bg_featmap:
    feat_upsample_list.{i}.layer_{1,2} (PixelShuffleUpsample, pixel_shuffle_upsample.py:25-29),
    feat_2_rgb_list.{0..n}, feat_layers.{i}; all Conv2d 1x1."""
    ch = [max(feat_nc // (2 ** i), min_feat) for i in range(n_blocks + 1)]
    shapes = OrderedDict()
    for i in range(n_blocks):
        shapes["feat_upsample_list.%d.layer_1" % i] = (2 * ch[i], ch[i])
        shapes["feat_upsample_list.%d.layer_2" % i] = (4 * ch[i], 2 * ch[i])
    for i in range(n_blocks + 1):
        shapes["feat_2_rgb_list.%d" % i] = (out_dim, ch[i])
    for i in range(n_blocks):
        shapes["feat_layers.%d" % i] = (ch[i + 1], ch[i])
    return shapes


def hash_renderer_params(seed: int = 0, feat_nc: int = FEAT_NC, n_blocks: int = 3, min_feat: int = 32,
                         weight_scale: float = 1.0):
    """NeuralRenderer parameters from PyTorch's Conv2d default init distribution
    (U(+-1/sqrt(fan_in)) for weight and bias), as a counter-based hash like hash_mlp_params."""
    out = OrderedDict()
    for name, (cout, cin) in renderer_param_shapes(feat_nc, n_blocks, min_feat).items():
        bound = 1.0 / math.sqrt(cin)
        u = hash_uniform(cout * cin, _key("renderer." + name + ".weight", seed))
        ub = hash_uniform(cout, _key("renderer." + name + ".bias", seed))
        out[name + ".weight"] = torch.from_numpy(((2.0 * u - 1.0) * bound * weight_scale).astype(np.float32).reshape(cout, cin, 1, 1))
        out[name + ".bias"] = torch.from_numpy(((2.0 * ub - 1.0) * bound).astype(np.float32))
    return out


def synth_featmap(batch: int, feat_nc: int, side: int, seed: int = 0):
    """A feature map shaped like the renderer's input: U(-0.5, 1.5) per element."""
    u = hash_uniform(batch * feat_nc * side * side, _key("featmap", seed))
    return torch.from_numpy((2.0 * u - 0.5).astype(np.float32).reshape(batch, feat_nc, side, side))


def synth_codes(batch: int, seed: int = 0):
    """shape_code ~ 0.5 N(0,1) [B,179], appea_code ~ 0.5 N(0,1) [B,127], gaze ~ U(-.5,.5) [B,2]."""
    n = SHAPE_DIMS + APPEA_DIMS
    u1 = hash_uniform(batch * n, _key("codes.u1", seed))
    u2 = hash_uniform(batch * n, _key("codes.u2", seed))
    g = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)   # Box-Muller
    g = (0.5 * g).astype(np.float32).reshape(batch, n)
    gaze = (hash_uniform(batch * GAZE_DIMS, _key("codes.gaze", seed)) - 0.5).astype(np.float32)
    return (torch.from_numpy(g[:, :SHAPE_DIMS].copy()),
            torch.from_numpy(g[:, SHAPE_DIMS:].copy()),
            torch.from_numpy(gaze.reshape(batch, GAZE_DIMS)))


def synth_jitter(batch: int, n_rays: int, n_samples: int, seed: int = 0) -> torch.Tensor:
    """t_rand ~ U[0,1) [B, N_r, N_p+1]: the stratified-jitter draw of model_utils.py:306."""
    u = hash_uniform(batch * n_rays * (n_samples + 1), _key("jitter", seed))
    return torch.from_numpy(u.astype(np.float32).reshape(batch, n_rays, n_samples + 1))


def synth_problem(side: int, batch: int = 1, camera: str = "frontal", seed: int = 0,
                  ray_subset=None):
    """Inputs of one hot-path call as a dict of CPU tensors."""
    xy = pixel_grid(side)
    if ray_subset is not None:
        xy = xy[:, :, ray_subset]
    xy = xy.expand(batch, -1, -1).contiguous()
    kinv = scaled_kinv(side).expand(batch, -1, -1).contiguous()
    rs, ts = [], []
    for b in range(batch):
        if camera == "frontal":
            r, t = frontal_camera()
        else:
            r, t = orbit_camera((int(camera) + 7 * b) % 45)
        rs.append(r)
        ts.append(t)
    shape, appea, gaze = synth_codes(batch, seed)
    return dict(xy=xy, R=torch.cat(rs, 0), T=torch.cat(ts, 0), Kinv=kinv,
                shape_code=shape, appea_code=appea, gaze=gaze)
