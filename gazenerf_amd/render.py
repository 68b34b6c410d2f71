"""The render op: GazeNeRF's volumetric hot path on MI355X, behind libgnr.so.

``render_two_stream`` replaces the span of the reference from ``self.sample_func(...)``
(models/gaze_nerf.py:231) through the two ``self.calc_color_func(...)`` calls
(models/gaze_nerf.py:157-162): GenSamplePoints -> Embedder -> latent concat ->
MLPforNeRF (face, eyes) -> CalcRayColor.  Argument meaning follows the reference:

    batch_xy [B,2,N_r]  R [B,3,3] c2w  T [B,3,1]  Kinv [B,3,3]
    shape_code [B,179]  gaze [B,2]  appea_code [B,127]
    face_params / eyes_params: the 24 tensors of one MLPforNeRF each, in ``PARAM_ORDER``
      (weights in Conv2d layout [out,in,1,1] or [out,in])
    t_rand None == reference ``disturb=False`` ("test"); a [B,N_r,N_p+1] tensor == "train" with
      exactly that stratified jitter (replaces torch.rand_like, utils/model_utils.py:306)

Gradients flow to R, T, shape_code, gaze, appea_code and all 48 parameter tensors through a
``torch.autograd.Function`` whose forward/backward are single calls into the C ABI.
PyTorch here is plumbing: it owns the device buffers and the stream.  All inputs must be CUDA
(ROCm) float32 tensors; anything else raises -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

from . import _lib, _torch_ext

PARAM_ORDER = tuple(
    ["FeaExt_module_%d.%s" % (i, k) for i in range(8) for k in ("weight", "bias")]
    + ["density_module.weight", "density_module.bias"]
    + ["RGB_layer_%d.%s" % (i, k) for i in range(3) for k in ("weight", "bias")])
N_PARAMS = len(PARAM_ORDER)   # 24


def params_to_list(params) -> list:
    """Accept a dict/state-dict (reference key names) or a 24-sequence in PARAM_ORDER."""
    if isinstance(params, dict):
        return [params[k] for k in PARAM_ORDER]
    params = list(params)
    if len(params) != N_PARAMS:
        raise ValueError("expected %d parameter tensors, got %d" % (N_PARAMS, len(params)))
    return params


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _check_tensor(name, t, shape=None):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on a CUDA/ROCm device (the render op has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))


def _weights_struct(plist: Sequence[torch.Tensor], cls=_lib.GnrWeights):
    w = cls()
    for i in range(8):
        w.fea_w[i] = plist[2 * i].data_ptr() if plist[2 * i] is not None else None
        w.fea_b[i] = plist[2 * i + 1].data_ptr() if plist[2 * i + 1] is not None else None
    w.density_w = plist[16].data_ptr() if plist[16] is not None else None
    w.density_b = plist[17].data_ptr() if plist[17] is not None else None
    for i in range(3):
        w.rgb_w[i] = plist[18 + 2 * i].data_ptr() if plist[18 + 2 * i] is not None else None
        w.rgb_b[i] = plist[19 + 2 * i].data_ptr() if plist[19 + 2 * i] is not None else None
    return w


def _expected_param_shapes(hidden, vp, appea, feat_nc, vd=0):
    shapes = [(hidden, vp), (hidden,)]
    for i in range(1, 8):
        shapes += [(hidden, hidden + vp if i == 5 else hidden), (hidden,)]
    shapes += [(1, hidden), (1,)]
    shapes += [(hidden, hidden), (hidden,), (hidden // 2, hidden + vd + appea), (hidden // 2,),
               (feat_nc, hidden // 2), (feat_nc,)]
    return shapes


def _prep_params(plist, hidden, vp, appea, feat_nc, tag, vd=0):
    out = []
    for name, t, shp in zip(PARAM_ORDER, plist, _expected_param_shapes(hidden, vp, appea, feat_nc, vd)):
        _check_tensor("%s.%s" % (tag, name), t)
        if t.dim() == 4:                       # Conv2d [out,in,1,1] == row-major [out,in]
            if t.shape[2:] != (1, 1):
                raise ValueError("%s.%s: only 1x1 kernels" % (tag, name))
            t = t.reshape(t.shape[0], t.shape[1])
        if tuple(t.shape) != shp:
            raise ValueError("%s.%s must have shape %s, got %s" % (tag, name, shp, tuple(t.shape)))
        out.append(t.contiguous())
    return out


class _Problem:
    """Validated, contiguous inputs + the ctypes GnrProblem that points at them."""

    def __init__(self, xy, R, T, Kinv, shape_code, gaze, appea_code, n_samples, world_z1, world_z2,
                 t_rand, z_edges, hidden, feat_nc, edges_follow_T=False, vd_dims=0, ray_bias=(None, None)):
        _check_tensor("batch_xy", xy)
        if xy.dim() != 3 or xy.shape[1] != 2:
            raise ValueError("batch_xy must be [B,2,N_r], got %s" % (tuple(xy.shape),))
        B, _, n_r = xy.shape
        _check_tensor("R", R, (B, 3, 3))
        _check_tensor("T", T, (B, 3, 1))
        _check_tensor("Kinv", Kinv, (B, 3, 3))
        _check_tensor("shape_code", shape_code)
        _check_tensor("gaze", gaze)
        _check_tensor("appea_code", appea_code)
        if shape_code.shape[0] != B or gaze.shape[0] != B or appea_code.shape[0] != B:
            raise ValueError("latent codes must have batch %d" % B)
        if t_rand is not None:
            _check_tensor("t_rand", t_rand, (B, n_r, n_samples + 1))
        if z_edges is not None:
            _check_tensor("z_edges", z_edges, (B, n_r, n_samples + 1))
        self.tensors = [t.contiguous() if t is not None else None
                        for t in (xy, R, T, Kinv, shape_code, gaze, appea_code, t_rand, z_edges)]
        xy, R, T, Kinv, shape_code, gaze, appea_code, t_rand, z_edges = self.tensors
        p = _lib.GnrProblem()
        p.batch, p.n_rays, p.n_samples = B, n_r, int(n_samples)
        p.hidden, p.feat_nc = int(hidden), int(feat_nc)
        p.shape_dims, p.gaze_dims, p.appea_dims = shape_code.shape[1], gaze.shape[1], appea_code.shape[1]
        p.world_z1, p.world_z2 = float(world_z1), float(world_z2)
        p.xy, p.R, p.T, p.Kinv = xy.data_ptr(), R.data_ptr(), T.data_ptr(), Kinv.data_ptr()
        p.shape_code, p.gaze, p.appea_code = shape_code.data_ptr(), gaze.data_ptr(), appea_code.data_ptr()
        p.t_rand = t_rand.data_ptr() if t_rand is not None else None
        p.z_edges = z_edges.data_ptr() if z_edges is not None else None
        p.edges_follow_T = 1 if (edges_follow_T and z_edges is not None) else 0
        # view-direction option: vd_dims columns of RGB_layer_1.weight are skipped by the kernels, their contribution
        # arrives as a per-ray bias [B, N_r, hidden/2] per weight set (include/gnr.h)
        p.vd_dims = int(vd_dims)
        self.ray_bias = []
        for s, rb in enumerate(ray_bias):
            if rb is not None:
                _check_tensor("ray_bias", rb, (B, n_r, int(hidden) // 2))
                rb = rb.contiguous()
                p.ray_bias[s] = rb.data_ptr()
            self.ray_bias.append(rb)
        self.uses_vd = p.vd_dims != 0 or any(rb is not None for rb in self.ray_bias)
        self.c = p
        self.B, self.n_r, self.n_p = B, n_r, int(n_samples)
        self.device = xy.device
        self.vp = 63 + p.shape_dims + p.gaze_dims


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _alloc_ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def _ext_call(fn, *args):
    """TORCH_CHECK failures of the C++ binding surface as the same exception class the ctypes binding raises."""
    try:
        return fn(*args)
    except torch.cuda.OutOfMemoryError:
        raise                          # an allocator failure is not an argument error: OOM handlers must still match
    except RuntimeError as e:
        raise _lib.GnrError(str(e).split("\n")[0]) from e


class PackedWeightCache:
    """Caller-owned inference workspace whose re-laid-out weights survive from call to call (GnrProblem.weights_packed).

    Every gnr_fwd re-packs 2 x 5.4 MB of weights into its workspace (12-54 us per weight set: 1-3 % of a 64 x 64-ray
    inference).  With a cache the op runs inference calls in the cache's workspace and skips the re-layout when
    nothing changed: same precision, same problem dimensions, and every parameter is the SAME tensor object at the
    SAME ``_version`` (torch bumps it on every in-place write, e.g. an optimizer step or ``load_state_dict``).  The
    cache holds references to the parameter tensors, so a freed-and-reallocated address cannot alias a key.
    Training calls (anything that needs gradients) never use it.  One cache per module and HIP stream: concurrent
    calls on two streams would share the workspace.

    The key also holds every parameter's ``data_ptr()``, so ``p.data = other`` (``Module._apply`` / ``.to()``, EMA
    swaps) misses.  What torch does NOT record is an in-place write through ``.data`` (``p.data.copy_()``,
    ``p.data[:] = ...``: same storage, ``_version`` unchanged) -- the reference initialises weights that way
    (models/mlp_nerf.py:61-75).  ``HotPathRenderer`` therefore clears its caches in ``train()``, ``_apply()`` and
    ``load_state_dict()``; a caller who edits ``.data`` of an eval-mode module in place must call ``clear()``."""

    def __init__(self):
        self.ws = None
        self.shape_key = None
        self.params = None
        self.versions = None
        self.hits = 0
        self.misses = 0

    def lookup(self, nbytes: int, shape_key, params):
        """(workspace, packed weights still valid)."""
        same_shape = self.ws is not None and self.shape_key == shape_key and self.ws.numel() >= nbytes
        hit = (same_shape and self.params is not None and len(self.params) == len(params) and
               all(a is b and b._version == v and b.data_ptr() == d
                   for a, b, (v, d) in zip(self.params, params, self.versions)))
        if not same_shape:
            # a fresh workspace holds no packed weights: forget the key NOW, so that a call that raises before store()
            # (e.g. an OOM while allocating outputs, swallowed by a trainer's try/except) cannot leave the old key
            # describing the new, uninitialised buffer
            self.shape_key = self.params = self.versions = None
            self.ws = _alloc_ws(max(int(nbytes), 256), shape_key[-1])
        self.hits += int(hit)
        self.misses += int(not hit)
        return self.ws, hit

    def store(self, shape_key, params):
        self.shape_key = shape_key
        self.params = list(params)
        self.versions = [(p._version, p.data_ptr()) for p in params]

    def clear(self):
        self.__init__()


def _run_forward(prob: _Problem, streams, save: bool, want_depth: bool, want_weights: bool, bf16x3: bool = False,
                 cache: Optional[PackedWeightCache] = None, cache_params=None):
    ext = _torch_ext.active()
    cws, packed, shape_key = None, False, None
    if cache is not None and not save:
        c = prob.c
        nb = _lib.load().gnr_workspace_bytes(C.byref(c), len(streams), _lib.WS_FWD)
        if nb == 0:
            _lib.check(1, _lib.load())
        shape_key = (bool(bf16x3), len(streams), c.batch, c.n_rays, c.n_samples, c.hidden, c.feat_nc, c.shape_dims,
                     c.gaze_dims, c.appea_dims, c.vd_dims, int(nb), prob.device)
        cws, packed = cache.lookup(nb, shape_key, cache_params)
        # the parameters the kernels read must be the cached tensors themselves, not contiguous copies of them
        packed = packed and all(a.data_ptr() == b.data_ptr() for a, b in zip(cache_params, [t for st in streams for t in st]))
    prob.c.weights_packed = 1 if packed else 0
    if ext is not None:                # C++ binding: device guard, current stream, allocation and checks in C++
        t = prob.tensors
        flat = _ext_call(ext.render_fwd, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], streams[0],
                              streams[1] if len(streams) > 1 else [], prob.n_p, prob.c.world_z1, prob.c.world_z2,
                              prob.c.hidden, prob.c.feat_nc, bool(save), bool(want_depth), bool(want_weights), bool(bf16x3),
                              bool(prob.c.edges_follow_T), cws, bool(packed), int(prob.c.vd_dims), prob.ray_bias[0],
                              prob.ray_bias[1])
        if cws is not None:
            cache.store(shape_key, cache_params)
        per = 2 + int(want_depth) + int(want_weights)
        res = []
        for s in range(len(streams)):
            o = flat[s * per:(s + 1) * per]
            res.append((o[0], o[1], o[2] if want_depth else None, o[-1] if want_weights else None))
        return res, flat[-1]
    lib = _lib.load()
    dev = prob.device
    n_streams = len(streams)
    feat_nc = prob.c.feat_nc
    kind = _lib.WS_FWD_SAVE if save else _lib.WS_FWD
    with torch.cuda.device(dev):
        nbytes = lib.gnr_workspace_bytes(C.byref(prob.c), n_streams, kind)
        if nbytes == 0:
            _lib.check(1, lib)
        ws = cws if cws is not None else _alloc_ws(nbytes, dev)
        outs = _lib.GnrOutputs()
        res = []
        for s in range(n_streams):
            feat = torch.empty(prob.B, feat_nc, prob.n_r, device=dev, dtype=torch.float32)
            bga = torch.empty(prob.B, 1, prob.n_r, device=dev, dtype=torch.float32)
            dep = torch.empty(prob.B, 1, prob.n_r, device=dev, dtype=torch.float32) if want_depth else None
            wts = (torch.empty(prob.B, 1, prob.n_r, prob.n_p, device=dev, dtype=torch.float32)
                   if want_weights else None)
            outs.feat[s], outs.bg_alpha[s] = feat.data_ptr(), bga.data_ptr()
            outs.depth[s] = dep.data_ptr() if dep is not None else None
            outs.weights[s] = wts.data_ptr() if wts is not None else None
            res.append((feat, bga, dep, wts))
        w0 = _weights_struct(streams[0])
        w1 = _weights_struct(streams[1]) if n_streams > 1 else None
        fwd = lib.gnr_fwd_bf16x3 if bf16x3 else lib.gnr_fwd
        rc = fwd(C.byref(prob.c), C.byref(w0), C.byref(w1) if w1 is not None else None,
                 C.byref(outs), 1 if save else 0, C.c_void_p(ws.data_ptr()), ws.numel(), _stream_ptr(dev))
        _lib.check(rc, lib)
        if cws is not None:
            cache.store(shape_key, cache_params)
    return res, ws


def _run_backward(prob: _Problem, streams, gout, saved_ws, bf16x3: bool):
    """One gnr_bwd call.  ``gout``: per stream (d feat [B,C,N_r] | None, d bg_alpha [B,1,N_r] | None), contiguous
    fp32.  Returns ([gR, gT, gshape, ggaze, gappea], [[24 parameter gradients] per stream], [d ray_bias per stream])."""
    ext = _torch_ext.active()
    if ext is not None:
        t = prob.tensors
        flat = _ext_call(ext.render_bwd, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], streams[0],
                              streams[1] if len(streams) > 1 else [], [g[0] for g in gout], [g[1] for g in gout], saved_ws,
                              prob.n_p, prob.c.world_z1, prob.c.world_z2, prob.c.hidden, prob.c.feat_nc, bool(bf16x3),
                              bool(prob.c.edges_follow_T), int(prob.c.vd_dims), prob.ray_bias[0], prob.ray_bias[1])
        return (list(flat[:5]), [list(flat[5 + 24 * s:5 + 24 * (s + 1)]) for s in range(len(streams))],
                [flat[-2], flat[-1]])
    lib = _lib.load()
    dev = prob.device
    n_streams = len(streams)
    dout = _lib.GnrOutputGrads()
    for s, (gf, ga) in enumerate(gout):
        dout.feat[s] = gf.data_ptr() if gf is not None else None
        dout.bg_alpha[s] = ga.data_ptr() if ga is not None else None
    B = prob.B
    gin = [torch.empty(B, 3, 3, device=dev), torch.empty(B, 3, 1, device=dev),
           torch.empty(B, prob.c.shape_dims, device=dev), torch.empty(B, prob.c.gaze_dims, device=dev),
           torch.empty(B, prob.c.appea_dims, device=dev)]
    din = _lib.GnrInputGrads()
    din.R, din.T, din.shape_code, din.gaze, din.appea_code = (t.data_ptr() for t in gin)
    grb = [None, None]
    for s, rb in enumerate(prob.ray_bias[:n_streams]):
        if rb is not None:
            grb[s] = torch.empty_like(rb)
            din.ray_bias[s] = grb[s].data_ptr()
    gparams = [[torch.empty_like(t) for t in streams[s]] for s in range(n_streams)]
    dw = [_weights_struct(g, _lib.GnrWeightGrads) for g in gparams]
    w = [_weights_struct(st) for st in streams]
    with torch.cuda.device(dev):
        nbytes = lib.gnr_workspace_bytes(C.byref(prob.c), n_streams, _lib.WS_BWD)
        scratch = _alloc_ws(max(nbytes, 256), dev)
        bwd = lib.gnr_bwd_bf16x3 if bf16x3 else lib.gnr_bwd
        rc = bwd(C.byref(prob.c), C.byref(w[0]), C.byref(w[1]) if n_streams > 1 else None,
                 C.byref(dout), C.byref(din), C.byref(dw[0]),
                 C.byref(dw[1]) if n_streams > 1 else None,
                 C.c_void_p(saved_ws.data_ptr()), saved_ws.numel(),
                 C.c_void_p(scratch.data_ptr()), scratch.numel(), _stream_ptr(dev))
        _lib.check(rc, lib)
    return gin, gparams, grb


# Saved activations cost ~31.5 KB per sample (both streams): 16.5 GB at cfg3 (2 x 4096 rays x 64), 540 GB for one
# 512 x 512-ray image.  Above this budget the op switches to ray tiles: the forward runs once without saving
# (the faster inference kernel), the backward re-runs forward-with-save tile by tile and accumulates -- the same
# recompute trade the reference would need torch.utils.checkpoint for.  GNR_WS_BUDGET_GB overrides the default.
DEFAULT_WS_BUDGET = int(float(os.environ.get("GNR_WS_BUDGET_GB", "96")) * (1 << 30))
_TILE_GRANULE = 256       # rays; tiles are multiples of this (a whole number of workgroups for any chunk count)


def plan_ray_tiles(prob: "_Problem", n_streams: int, budget_bytes: Optional[int] = None,
                   ray_tile: Optional[int] = None):
    """None == the saved workspace of the whole problem fits the budget (no tiling); otherwise the tile size in
    rays.  An explicit ``ray_tile`` forces tiling with that size."""
    if ray_tile is not None:
        if ray_tile < 1:
            raise ValueError("ray_tile must be >= 1")
        return None if ray_tile >= prob.n_r else int(ray_tile)
    lib = _lib.load()
    budget = DEFAULT_WS_BUDGET if budget_bytes is None else int(budget_bytes)
    full = lib.gnr_workspace_bytes(C.byref(prob.c), n_streams, _lib.WS_FWD_SAVE)
    if full == 0:
        _lib.check(1, lib)
    if full <= budget:
        return None
    per_ray = full / prob.n_r                         # the workspace is linear in rays up to fixed terms
    tile = int(budget / per_ray) // _TILE_GRANULE * _TILE_GRANULE
    return max(_TILE_GRANULE, min(tile, prob.n_r))


class _RenderFn(torch.autograd.Function):
    """forward = gnr_fwd, backward = gnr_bwd.  Inputs: 5 differentiable problem tensors, then
    24 * n_streams parameter tensors; non-differentiable context travels in ``cfg``.

    Every tensor the backward reads goes through ``ctx.save_for_backward`` (so autograd's version check catches
    an in-place update between forward and backward -- e.g. an optimizer step -- and frees / retains them with the
    graph), including the activation workspace of the un-tiled mode."""

    @staticmethod
    def forward(ctx, cfg, R, T, shape_code, gaze, appea_code, rb0, rb1, *flat_params):
        n_streams = cfg["n_streams"]
        vd = cfg.get("vd_dims", 0)
        prob = _Problem(cfg["xy"], R, T, cfg["Kinv"], shape_code, gaze, appea_code, cfg["n_samples"],
                        cfg["world_z1"], cfg["world_z2"], cfg["t_rand"], cfg["z_edges"],
                        cfg["hidden"], cfg["feat_nc"], cfg.get("edges_follow_T", False), vd, (rb0, rb1))
        streams = [_prep_params(flat_params[24 * s:24 * (s + 1)], cfg["hidden"], prob.vp,
                                prob.c.appea_dims, cfg["feat_nc"], "stream%d" % s, vd)
                   for s in range(n_streams)]
        need_grad = any(ctx.needs_input_grad[1:])
        bf16x3 = cfg.get("precision", "fp32") == "bf16x3"
        tile = plan_ray_tiles(prob, n_streams, cfg.get("ws_budget_bytes"), cfg.get("ray_tile")) if need_grad else None
        save = need_grad and tile is None
        res, ws = _run_forward(prob, streams, save, cfg["want_depth"], cfg["want_weights"], bf16x3,
                               cache=None if need_grad else cfg.get("weight_cache"), cache_params=flat_params)
        ctx.cfg = {k: v for k, v in cfg.items() if not torch.is_tensor(v) and k != "weight_cache"}
        ctx.tile, ctx.need_grad = tile, need_grad
        ctx.param_shapes = [tuple(t.shape) for t in flat_params]
        if need_grad:
            ctx.save_for_backward(R, T, shape_code, gaze, appea_code, *flat_params, cfg["xy"], cfg["Kinv"],
                                  cfg["t_rand"], cfg["z_edges"], ws if save else None, rb0, rb1)
        outs = []
        nondiff = []
        for feat, bga, dep, wts in res:
            outs += [feat, bga]
            for extra in (dep, wts):
                if extra is not None:
                    outs.append(extra)
                    nondiff.append(extra)
        ctx.mark_non_differentiable(*nondiff)
        ctx.out_layout = [(dep is not None, wts is not None) for _, _, dep, wts in res]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if not ctx.need_grad:
            raise RuntimeError("render_two_stream: backward called but no input required a gradient in forward")
        cfg = ctx.cfg
        saved = ctx.saved_tensors
        n_streams = cfg["n_streams"]
        R, T, shape_code, gaze, appea_code = saved[:5]
        flat_params = saved[5:5 + 24 * n_streams]
        xy, Kinv, t_rand, z_edges, ws, rb0, rb1 = saved[5 + 24 * n_streams:]
        bf16x3 = cfg.get("precision", "fp32") == "bf16x3"
        vd = cfg.get("vd_dims", 0)

        def problem(sl=None):
            cut = (lambda t, dim: None if t is None else t[(slice(None),) * dim + (sl,)].contiguous()) if sl else (lambda t, dim: t)
            return _Problem(cut(xy, 2), R, T, Kinv, shape_code, gaze, appea_code, cfg["n_samples"], cfg["world_z1"],
                            cfg["world_z2"], cut(t_rand, 1), cut(z_edges, 1), cfg["hidden"], cfg["feat_nc"],
                            cfg.get("edges_follow_T", False), vd, (cut(rb0, 1), cut(rb1, 1)))

        streams = None
        gout_full = []
        gi = 0
        for has_d, has_w in ctx.out_layout:
            gf, ga = grads[gi], grads[gi + 1]
            gi += 2 + int(has_d) + int(has_w)
            gout_full.append(tuple(None if g is None else g.contiguous().float() for g in (gf, ga)))

        if ctx.tile is None:
            prob = problem()
            streams = [_prep_params(flat_params[24 * s:24 * (s + 1)], cfg["hidden"], prob.vp, prob.c.appea_dims,
                                    cfg["feat_nc"], "stream%d" % s, vd) for s in range(n_streams)]
            gin, gparams, grb = _run_backward(prob, streams, gout_full, ws, bf16x3)
        else:
            gin = gparams = None
            grb = [None if rb is None else torch.empty_like(rb) for rb in (rb0, rb1)]
            n_r = xy.shape[2]
            for r0 in range(0, n_r, ctx.tile):
                sl = slice(r0, min(n_r, r0 + ctx.tile))
                prob = problem(sl)
                if streams is None:
                    streams = [_prep_params(flat_params[24 * s:24 * (s + 1)], cfg["hidden"], prob.vp,
                                            prob.c.appea_dims, cfg["feat_nc"], "stream%d" % s, vd) for s in range(n_streams)]
                _, tws = _run_forward(prob, streams, True, False, False, bf16x3)        # recompute with save
                gout = [tuple(None if g is None else g[:, :, sl].contiguous() for g in pair) for pair in gout_full]
                tin, tpar, trb = _run_backward(prob, streams, gout, tws, bf16x3)
                del tws
                for full, part in zip(grb, trb):       # a ray's bias gradient belongs to exactly one tile
                    if full is not None:
                        full[:, sl] = part
                if gin is None:
                    gin, gparams = tin, tpar
                else:                                   # fixed tile order: deterministic accumulation
                    torch._foreach_add_(gin, tin)
                    for a, b in zip(gparams, tpar):
                        torch._foreach_add_(a, b)
        flat = []
        for s in range(n_streams):
            for g, shp in zip(gparams[s], ctx.param_shapes[24 * s:24 * (s + 1)]):
                flat.append(g.reshape(shp))
        return (None, *gin, grb[0], grb[1], *flat)


def render_two_stream(batch_xy, R, T, Kinv, shape_code, gaze, appea_code, face_params, eyes_params=None,
                      *, n_samples: int, world_z1: float = 2.5, world_z2: float = -3.5,
                      t_rand: Optional[torch.Tensor] = None, z_edges: Optional[torch.Tensor] = None,
                      return_depth: bool = False, return_weights: bool = False,
                      hidden: int = 384, feat_nc: int = 258, precision: str = "fp32",
                      edges_follow_T: bool = False, ray_tile: Optional[int] = None,
                      ws_budget_bytes: Optional[int] = None, weight_cache: Optional["PackedWeightCache"] = None,
                      vd_dims: int = 0, ray_bias_face: Optional[torch.Tensor] = None,
                      ray_bias_eyes: Optional[torch.Tensor] = None):
    """Run the hot path.  Returns a dict with feat_face [B,feat_nc,N_r], bg_alpha_face [B,1,N_r]
    (and *_eyes when ``eyes_params`` is given; depth_* / w_* [B,1,N_r,N_p] on request).

    ``eyes_params=None`` evaluates a single MLP -- the hierarchical fine pass with the third
    network (models/gaze_nerf.py:102-108); its results come back under the "face" keys.

    ``precision="bf16x3"`` runs the dense layers (forward and the dgrad chain of the backward) on bf16
    MFMA with a 3-term hi/lo split of both operands; it agrees with the default exact-fp32 path to that
    path's own rounding noise (see gnr_fwd_bf16x3 / gnr_bwd_bf16x3 in include/gnr.h).

    ``edges_follow_T`` (with ``z_edges``): the edges shift 1:1 with T_z, as FineSample's merged edges do in the
    reference (only the weights are detached, utils/model_utils.py:418, 455-476), so dL/dT is the plane sweep's.

    Training calls whose saved activations exceed ``ws_budget_bytes`` (default 96 GB, env GNR_WS_BUDGET_GB; one
    512x512-ray image would need 540 GB) run in ray tiles inside the op: forward once without saving, backward
    recomputes forward-with-save per tile and accumulates the gradients in a fixed order.  ``ray_tile`` forces a
    tile size.  The reference makes one forward call for the whole image (models/gaze_nerf.py:211-320); so does
    the caller of this op.

    ``weight_cache`` (a ``PackedWeightCache`` the caller keeps): inference calls run in its workspace and skip the
    weight re-layout while the parameters are unchanged.

    ``vd_dims`` / ``ray_bias_*``: the reference's view-direction option (``include_vd``, models/gaze_nerf.py:70-80,
    140).  ``RGB_layer_1.weight`` is then [H/2, H + vd_dims + appea]; the kernels skip the vd_dims columns and add
    ``ray_bias_* [B, N_r, H/2]`` (differentiable) to that layer's bias for every sample of a ray -- the direction is
    constant along a ray, so the caller folds ``W[:, H:H+vd_dims] @ embed(dir)`` per ray (``HotPathRenderer`` does).
    """
    if precision not in ("fp32", "bf16x3"):
        raise ValueError("precision must be 'fp32' or 'bf16x3'")
    streams = [params_to_list(face_params)]
    if eyes_params is not None:
        streams.append(params_to_list(eyes_params))
    cfg = dict(xy=batch_xy, Kinv=Kinv, n_samples=int(n_samples), world_z1=world_z1, world_z2=world_z2,
               t_rand=t_rand, z_edges=z_edges, hidden=hidden, feat_nc=feat_nc, n_streams=len(streams),
               want_depth=return_depth, want_weights=return_weights, precision=precision,
               edges_follow_T=bool(edges_follow_T), ray_tile=ray_tile, ws_budget_bytes=ws_budget_bytes,
               weight_cache=weight_cache, vd_dims=int(vd_dims))
    flat = [t for st in streams for t in st]
    outs = _RenderFn.apply(cfg, R, T, shape_code, gaze, appea_code, ray_bias_face, ray_bias_eyes, *flat)
    res: Dict[str, torch.Tensor] = {}
    it = iter(outs)
    for tag in ("face", "eyes")[:len(streams)]:
        res["feat_" + tag] = next(it)
        res["bg_alpha_" + tag] = next(it)
        if return_depth:
            res["depth_" + tag] = next(it)
        if return_weights:
            res["w_" + tag] = next(it)
    return res


def render_two_stream_tiled(batch_xy, R, T, Kinv, shape_code, gaze, appea_code, face_params, eyes_params=None, *,
                            loss_fn, n_samples: int, ray_tile: Optional[int] = None, t_rand: Optional[torch.Tensor] = None,
                            z_edges: Optional[torch.Tensor] = None, ray_bias_face: Optional[torch.Tensor] = None,
                            ray_bias_eyes: Optional[torch.Tensor] = None, return_outputs: bool = False, **kw):
    """A TRAINING step over ray tiles at 3/3 of the FLOPs -- forward-with-save, loss, backward per tile, nothing recomputed.

    ``render_two_stream`` cannot see the loss: when the saved activations of a call exceed the workspace budget (one
    512 x 512-ray image: 540 GB) it has to run an inference forward first and then recompute forward-with-save per tile
    in its backward, 4/3 of the FLOPs.  A loss that is a SUM OVER RAYS does not need the whole image before the first
    backward -- the reference's image losses are such sums (masked L1 means over pixels, gazenerf_loss.py:438-468; the
    caller shape being replaced: models/gaze_nerf.py:318-351 + trainer/gazenerf_trainer.py:487-528).  So:

        for each tile of rays, in order:
            out  = render_two_stream(tile)                 # saves the tile's activations
            loss = loss_fn(out, sl)                        # sl = slice of the ray axis; out[k] is [B, C, len(sl)]
            loss.backward()                                # frees them; .grad of every leaf accumulates in tile order

    ``loss_fn(out, sl) -> scalar`` must return this tile's SHARE of the total loss (normalise by the total ray count,
    not the tile's, if the loss is a mean); any other tensors it touches (targets, masks) are sliced with ``sl`` by the
    caller.  ``loss_fn`` must not run a shared NON-LEAF graph on every tile (e.g. a target produced by another network
    with ``requires_grad``): the first tile's ``backward()`` frees that graph and the second raises "backward through the
    graph a second time" -- detach such tensors, or make them leaves, before the call.  Gradients end up where ``loss.backward()`` would put them -- ``.grad`` of leaf inputs, upstream of non-leaf
    ones -- summed over the tiles in a fixed order (deterministic).  Returns ``(total_loss_detached, outputs or None)``; ``return_outputs=True`` also returns the
    detached outputs of the whole image, concatenated along the ray axis.

    ``ray_tile=None``: the largest tile whose saved activations fit the workspace budget (``ws_budget_bytes`` / 96 GB).
    Other keyword arguments are ``render_two_stream``'s."""
    n_r = batch_xy.shape[2]
    if ray_tile is None:
        streams = 1 if eyes_params is None else 2
        # the problem the op itself would plan with (the view-direction option adds the per-ray bias and its gradient to
        # the saved workspace)
        dummy = _Problem(batch_xy, R, T, Kinv, shape_code, gaze, appea_code, n_samples, kw.get("world_z1", 2.5),
                         kw.get("world_z2", -3.5), None, None, kw.get("hidden", 384), kw.get("feat_nc", 258),
                         vd_dims=kw.get("vd_dims", 0), ray_bias=(ray_bias_face, ray_bias_eyes))
        ray_tile = plan_ray_tiles(dummy, streams, kw.get("ws_budget_bytes")) or n_r
    if ray_tile < 1:
        raise ValueError("ray_tile must be >= 1")
    kw = dict(kw)
    kw.pop("ws_budget_bytes", None)
    cut = lambda t, sl: None if t is None else t[:, sl]

    # The op's differentiable inputs enter every tile as detached PROXIES: a tile's backward leaves its gradients in
    # proxy.grad (set, not added), they are summed over the tiles with one multi-tensor add per tile, and the totals are
    # pushed into the real inputs once at the end -- leaves accumulate into .grad as usual, non-leaf inputs (a rotation
    # built from Euler angles, codes built from offsets) propagate upstream ONCE instead of once per tile.  With
    # loss.backward() on the real inputs autograd issued one small add kernel per tensor and tile: 53 x 15 per 512 x 512
    # image, 0.4 % of the step.  Anything else loss_fn touches gets its gradients from the per-tile backward directly.
    def proxy(t):
        return t.detach().requires_grad_(True) if (torch.is_tensor(t) and t.requires_grad) else t

    def proxy_params(ps):
        if ps is None:
            return None, []
        if isinstance(ps, dict):
            d = {k: proxy(v) for k, v in ps.items()}
            return d, [(ps[k], d[k]) for k in ps if d[k] is not ps[k]]
        lst = [proxy(v) for v in ps]
        return lst, [(a, b) for a, b in zip(ps, lst) if b is not a]

    real = [R, T, shape_code, gaze, appea_code, ray_bias_face, ray_bias_eyes]
    prox = [proxy(t) for t in real]
    pairs = [(a, b) for a, b in zip(real, prox) if b is not a]
    fpx, fpairs = proxy_params(face_params)
    epx, epairs = proxy_params(eyes_params)
    pairs += fpairs + epairs
    acc = None
    total = None
    pieces = {} if return_outputs else None
    for r0 in range(0, n_r, ray_tile):
        sl = slice(r0, min(n_r, r0 + ray_tile))
        # ray_tile >= the tile's rays: the op itself never tiles (and never recomputes) inside a tile
        out = render_two_stream(batch_xy[:, :, sl], prox[0], prox[1], Kinv, prox[2], prox[3], prox[4], fpx, epx,
                                n_samples=n_samples, t_rand=cut(t_rand, sl), z_edges=cut(z_edges, sl),
                                ray_bias_face=cut(prox[5], sl), ray_bias_eyes=cut(prox[6], sl),
                                ray_tile=sl.stop - sl.start, **kw)
        loss = loss_fn(out, sl)
        if loss.dim() != 0:
            raise ValueError("loss_fn must return a scalar (this tile's share of the total loss)")
        loss.backward()
        total = loss.detach() if total is None else total + loss.detach()
        if pairs:
            got = [px.grad for _, px in pairs]
            for _, px in pairs:
                px.grad = None
            if acc is None:                      # the first tile's gradient tensors become the accumulators
                acc = [g if g is not None else None for g in got]
            else:
                idx = [i for i, g in enumerate(got) if g is not None and acc[i] is not None]
                if idx:
                    torch._foreach_add_([acc[i] for i in idx], [got[i] for i in idx])        # fixed tile order: deterministic
                for i, g in enumerate(got):
                    if acc[i] is None and g is not None:
                        acc[i] = g
        if pieces is not None:
            for k, v in out.items():
                pieces.setdefault(k, []).append(v.detach())
        del out, loss
    if pairs and acc is not None:
        keep = [(t, g) for (t, _), g in zip(pairs, acc) if g is not None]
        if keep:
            torch.autograd.backward([t for t, _ in keep], [g for _, g in keep])
    outs = None
    if pieces is not None:
        outs = {k: torch.cat(v, dim=-2 if k.startswith("w_") else -1) for k, v in pieces.items()}
    return total, outs


def sample_zvals(batch_xy, R, T, Kinv, *, n_samples: int, world_z1: float = 2.5, world_z2: float = -3.5,
                 t_rand=None, z_edges=None):
    """Left sample edges [B,1,N_r,N_p] -- ``fg_sample_dict["zvals"]`` (utils/model_utils.py:312-313)."""
    lib = _lib.load()
    B = batch_xy.shape[0]
    dummy = torch.zeros(B, 1, device=batch_xy.device)
    prob = _Problem(batch_xy, R, T, Kinv, dummy, dummy, dummy, n_samples, world_z1, world_z2, t_rand,
                    z_edges, 384, 258)
    out = torch.empty(B, 1, prob.n_r, prob.n_p, device=prob.device)
    with torch.cuda.device(prob.device):
        _lib.check(lib.gnr_sample_zvals(C.byref(prob.c), C.c_void_p(out.data_ptr()), _stream_ptr(prob.device)), lib)
    return out


def importance_resample(w_face, zvals, *, n_fine: int, u: Optional[torch.Tensor] = None, validate: bool = False):
    """FineSample.forward (utils/model_utils.py:413-490): coarse weights [B,1,N_r,N_c] + coarse left
    edges [B,1,N_r,N_c] -> sorted merged edges [B,N_r,N_c+n_fine+1], to be passed back as
    ``z_edges`` with ``n_samples = N_c + n_fine``.  ``u`` [B*N_r, n_fine+1] replaces torch.rand
    (``disturb=True``); None == the deterministic linspace of ``disturb=False``.  No gradient
    (the reference detaches the weights, model_utils.py:418).

    Precondition (include/gnr.h): every row of ``zvals`` is ascending -- what ``sample_zvals`` / the plane sweep gives
    for ``world_z1 > world_z2`` (the reference's 2.5 / -3.5).  The kernel merges two sorted lists where the reference
    calls ``torch.sort``; ``validate=True`` checks the rows first (one host synchronisation) and raises otherwise."""
    lib = _lib.load()
    w = w_face.detach()
    _check_tensor("w_face", w)
    _check_tensor("zvals", zvals, tuple(w.shape))
    B, _, n_r, nc = w.shape
    w = w.contiguous()
    z = zvals.detach().contiguous()
    if validate and nc > 1 and not bool((z[..., 1:] >= z[..., :-1]).all()):
        raise ValueError("importance_resample: zvals must be ascending along the sample axis (world_z1 > world_z2); "
                         "the merge-based kernel has no general sort")
    if u is not None:
        _check_tensor("u", u, (B * n_r, n_fine + 1))
        u = u.contiguous()
    out = torch.empty(B, n_r, nc + n_fine + 1, device=w.device)
    with torch.cuda.device(w.device):
        rc = lib.gnr_resample(C.c_void_p(w.data_ptr()), C.c_void_p(z.data_ptr()), _ptr(u), B * n_r, nc,
                              int(n_fine), C.c_void_p(out.data_ptr()), _stream_ptr(w.device))
        _lib.check(rc, lib)
    return out
