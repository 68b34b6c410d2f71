"""The 2-D upsampler after the hot path: NeuralRenderer on MI355X.

``neural_render`` replaces ``NeuralRenderer.forward`` (models/neural_renderer.py:100-113) including
``PixelShuffleUpsample`` (models/pixel_shuffle_upsample.py:33-42) and ``Blur`` (:7-16) by the HIP kernels
behind ``gnr_upsample_fwd`` / ``gnr_upsample_bwd``; ``NeuralRendererAMD`` is an ``nn.Module`` with the
reference's parameter names and shapes (``feat_upsample_list.i.layer_{1,2}``, ``feat_2_rgb_list.i``,
``feat_layers.i``, ``bg_featmap``) so ``check_dict["net"]`` entries under ``neural_render.*`` load into it.
CUDA (ROCm) float32 tensors only; no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import torch
from torch import nn

from . import _lib
from .render import _check_tensor, _stream_ptr


def renderer_param_names(n_blocks: int):
    names = []
    for i in range(n_blocks):
        names += ["feat_upsample_list.%d.layer_1" % i, "feat_upsample_list.%d.layer_2" % i]
    names += ["feat_2_rgb_list.%d" % i for i in range(n_blocks + 1)]
    names += ["feat_layers.%d" % i for i in range(n_blocks)]
    return [n + s for n in names for s in (".weight", ".bias")]


def _channels(feat_nc, n_blocks, min_feat):
    return [max(feat_nc // (2 ** i), min_feat) for i in range(n_blocks + 1)]


def _weights_struct(params: Dict[str, torch.Tensor], n_blocks, cls=_lib.GnrUpsampleWeights):
    w = cls()
    get = lambda k: params[k].data_ptr() if params.get(k) is not None else None
    for i in range(n_blocks):
        w.up1_w[i], w.up1_b[i] = get("feat_upsample_list.%d.layer_1.weight" % i), get("feat_upsample_list.%d.layer_1.bias" % i)
        w.up2_w[i], w.up2_b[i] = get("feat_upsample_list.%d.layer_2.weight" % i), get("feat_upsample_list.%d.layer_2.bias" % i)
        w.feat_w[i], w.feat_b[i] = get("feat_layers.%d.weight" % i), get("feat_layers.%d.bias" % i)
    for i in range(n_blocks + 1):
        w.rgb_w[i], w.rgb_b[i] = get("feat_2_rgb_list.%d.weight" % i), get("feat_2_rgb_list.%d.bias" % i)
    return w


def _expected_shapes(feat_nc, n_blocks, min_feat):
    ch = _channels(feat_nc, n_blocks, min_feat)
    shp = {}
    for i in range(n_blocks):
        shp["feat_upsample_list.%d.layer_1" % i] = (2 * ch[i], ch[i])
        shp["feat_upsample_list.%d.layer_2" % i] = (4 * ch[i], 2 * ch[i])
        shp["feat_layers.%d" % i] = (ch[i + 1], ch[i])
    for i in range(n_blocks + 1):
        shp["feat_2_rgb_list.%d" % i] = (3, ch[i])
    return shp


def _prep_upsample(cfg, x, flat):
    """Validated contiguous inputs + the ctypes problem pointing at them."""
    n_blocks, min_feat, final = cfg["n_blocks"], cfg["min_feat"], cfg["final_actvn"]
    names = renderer_param_names(n_blocks)
    _check_tensor("x", x)
    if x.dim() != 4 or x.shape[2] != x.shape[3]:
        raise ValueError("x must be [B, C, S, S], got %s" % (tuple(x.shape),))
    B, Cn, S, _ = x.shape
    shapes = _expected_shapes(Cn, n_blocks, min_feat)
    params = {}
    for name, t in zip(names, flat):
        _check_tensor(name, t)
        base = name.rsplit(".", 1)[0]
        want = shapes[base] if name.endswith(".weight") else (shapes[base][0],)
        t2 = t.reshape(t.shape[0], -1) if name.endswith(".weight") else t
        if tuple(t2.shape) != want:
            raise ValueError("%s must have shape %s (+[1,1]), got %s" % (name, want, tuple(t.shape)))
        params[name] = t2.contiguous()
    xc = x.contiguous()
    p = _lib.GnrUpsampleProblem()
    p.batch, p.feat_nc, p.featmap_size, p.n_blocks, p.min_feat, p.final_sigmoid = B, Cn, S, n_blocks, min_feat, int(final)
    p.x = xc.data_ptr()
    return p, xc, params, names


class _UpsampleFn(torch.autograd.Function):
    """Inputs, parameters and the activation workspace travel through ``ctx.save_for_backward`` (autograd's
    version check then catches an in-place parameter update between forward and backward)."""

    @staticmethod
    def forward(ctx, cfg, x, *flat):
        lib = _lib.load()
        p, xc, params, names = _prep_upsample(cfg, x, flat)
        B, S, n_blocks = p.batch, p.featmap_size, p.n_blocks
        dev = x.device
        with torch.cuda.device(dev):
            nbytes = lib.gnr_upsample_workspace_bytes(C.byref(p), _lib.UP_WS_FWD)
            if nbytes == 0:
                _lib.check(1, lib)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            img = torch.empty(B, 3, S << n_blocks, S << n_blocks, device=dev, dtype=torch.float32)
            w = _weights_struct(params, n_blocks)
            rc = lib.gnr_upsample_fwd(C.byref(p), C.byref(w), C.c_void_p(img.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                      _stream_ptr(dev))
            _lib.check(rc, lib)
        ctx.cfg = cfg
        ctx.need_grad = any(ctx.needs_input_grad[1:])
        if ctx.need_grad:
            ctx.save_for_backward(x, ws, *flat)
        return img

    @staticmethod
    def backward(ctx, g_img):
        lib = _lib.load()
        if not ctx.need_grad:
            raise RuntimeError("neural_render: backward called but no input required a gradient in forward")
        x, ws = ctx.saved_tensors[:2]
        flat = ctx.saved_tensors[2:]
        p, xc, params, names = _prep_upsample(ctx.cfg, x, flat)
        n_blocks = ctx.cfg["n_blocks"]
        dev = xc.device
        g = g_img.contiguous()
        # a NULL gradient pointer == not wanted (include/gnr.h): a frozen renderer (latent-code fitting, the reference's
        # optimize_gaze_direction, gazenerf_trainer.py:1152-1164) skips every weight-gradient GEMM, frozen inputs skip d_x
        want = ctx.needs_input_grad[1:]
        dx = torch.empty_like(xc) if want[0] else None
        grads = {k: torch.empty_like(v) for (k, v), need in zip(params.items(), want[1:]) if need}
        with torch.cuda.device(dev):
            nbytes = lib.gnr_upsample_workspace_bytes(C.byref(p), _lib.UP_WS_BWD)
            scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            w = _weights_struct(params, n_blocks)
            dw = _weights_struct(grads, n_blocks, _lib.GnrUpsampleWeightGrads)
            rc = lib.gnr_upsample_bwd(C.byref(p), C.byref(w), C.c_void_p(g.data_ptr()),
                                      C.c_void_p(dx.data_ptr()) if dx is not None else None,
                                      C.byref(dw) if grads else None,
                                      C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(scratch.data_ptr()),
                                      scratch.numel(), _stream_ptr(dev))
            _lib.check(rc, lib)
        return (None, dx.reshape(x.shape) if dx is not None else None) + \
            tuple(grads[n].reshape(t.shape) if n in grads else None for n, t in zip(names, flat))


def neural_render(x, params: Dict[str, torch.Tensor], n_blocks: int = 3, min_feat: int = 32, final_actvn: bool = True):
    """x [B, feat_nc, S, S] -> image [B, 3, S*2^n, S*2^n].  ``params``: the reference's NeuralRenderer state-dict
    entries (weights [out,in,1,1] or [out,in]); gradients flow to x and to every parameter."""
    names = renderer_param_names(n_blocks)
    cfg = dict(n_blocks=int(n_blocks), min_feat=int(min_feat), final_actvn=bool(final_actvn))
    return _UpsampleFn.apply(cfg, x, *[params[n] for n in names])


class GraphedUpsample:
    """Inference forward of the upsampler replayed from a HIP graph (round 4).

    One ``gnr_upsample_fwd`` call is 20 kernel launches; at B = 1 (the reference's render loops,
    utils/render_utils.py:199-219) 110 of its 512 us were the gaps between them.  The library makes no host
    synchronisation and takes every buffer from the caller, so the call captures as is: static input, workspace and image
    buffers, one ``torch.cuda.CUDAGraph`` per (input shape, device, parameter storage).  The weight re-layout
    (``conv16_pack_kernel``) is INSIDE the graph, so in-place parameter updates are seen by the next replay; a parameter
    that moves to other storage (``.to()``, ``load_state_dict`` into new tensors, ``p.data = ...``) changes the key and the
    forward is captured again.  Inference only: nothing is saved for a backward."""

    def __init__(self, max_entries: int = 4):
        self.entries = {}
        self.max_entries = max_entries
        self.captures = 0
        self.replays = 0

    def clear(self):
        self.entries.clear()

    # the cache holds CUDAGraph objects and ctypes structs of device pointers: a copy / pickle of the owning module starts
    # with an empty cache (and captures again on its first inference call)
    def __deepcopy__(self, memo):
        return GraphedUpsample(self.max_entries)

    def __getstate__(self):
        return {"max_entries": self.max_entries}

    def __setstate__(self, state):
        self.__init__(state.get("max_entries", 4))

    def __call__(self, x, params: Dict[str, torch.Tensor], n_blocks: int, min_feat: int, final_actvn: bool, copy_output: bool = True):
        names = renderer_param_names(n_blocks)
        flat = [params[n] for n in names]
        key = (tuple(x.shape), x.device, n_blocks, min_feat, bool(final_actvn)) + tuple(t.data_ptr() for t in flat) + \
            tuple(tuple(t.shape) for t in flat)
        ent = self.entries.get(key)
        if ent is None:
            ent = self._capture(x, flat, dict(n_blocks=int(n_blocks), min_feat=int(min_feat), final_actvn=bool(final_actvn)))
            if len(self.entries) >= self.max_entries:
                self.entries.pop(next(iter(self.entries)))
            self.entries[key] = ent
        graph, x_static, img_static = ent[:3]
        x_static.copy_(x)
        graph.replay()
        self.replays += 1
        return img_static.clone() if copy_output else img_static

    def _capture(self, x, flat, cfg):
        lib = _lib.load()
        dev = x.device
        # the static buffers outlive this call: allocated outside inference mode, so that a later replay under plain
        # torch.no_grad() may still copy_ into them (an inference tensor refuses in-place updates outside inference mode)
        with torch.inference_mode(False), torch.no_grad():
            return self._capture_impl(lib, dev, x, flat, cfg)

    def _capture_impl(self, lib, dev, x, flat, cfg):
        x_static = torch.empty(x.shape, dtype=x.dtype, device=dev)
        x_static.copy_(x)
        p, xc, params, names = _prep_upsample(cfg, x_static, flat)
        for n, t, f in zip(names, [params[n] for n in names], flat):
            if t.data_ptr() != f.data_ptr():
                raise ValueError("GraphedUpsample: parameter %s is not contiguous; the graph would read a temporary copy" % n)
        B, S, n_blocks = p.batch, p.featmap_size, p.n_blocks
        with torch.cuda.device(dev):
            nbytes = lib.gnr_upsample_workspace_bytes(C.byref(p), _lib.UP_WS_FWD)
            if nbytes == 0:
                _lib.check(1, lib)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            img = torch.empty(B, 3, S << n_blocks, S << n_blocks, device=dev, dtype=torch.float32)
            w = _weights_struct(params, n_blocks)

            def call():
                _lib.check(lib.gnr_upsample_fwd(C.byref(p), C.byref(w), C.c_void_p(img.data_ptr()), C.c_void_p(ws.data_ptr()),
                                                ws.numel(), _stream_ptr(dev)), lib)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                call()                                  # warm-up outside the capture (module load, argument errors)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                call()
        self.captures += 1
        return graph, x_static, img, ws, params, p, w      # the graph's kernels point into all of these: keep them alive


class _BlurBuf(nn.Module):
    """Holds Blur's registered buffer ``f = [1, 2, 1]`` (pixel_shuffle_upsample.py:8-11) so that the
    reference's state-dict keys ``*.blur_layer.f`` / ``rgb_upsample.1.f`` exist here too (strict loading)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("f", torch.tensor([1.0, 2.0, 1.0]))


class _PSU(nn.Module):
    """Parameter holder named like PixelShuffleUpsample (pixel_shuffle_upsample.py:25-29)."""

    def __init__(self, c):
        super().__init__()
        self.layer_1 = nn.Conv2d(c, 2 * c, 1, 1, padding=0)
        self.layer_2 = nn.Conv2d(2 * c, 4 * c, 1, 1, padding=0)
        self.blur_layer = _BlurBuf()


class NeuralRendererAMD(nn.Module):
    """Drop-in for models/neural_renderer.py:NeuralRenderer (same constructor arguments, parameter names,
    shapes and initialisation; ``forward`` runs on the HIP kernels)."""

    def __init__(self, bg_type="white", feat_nc=258, out_dim=3, final_actvn=True, min_feat=32, featmap_size=64,
                 img_size=512, graph_inference=True, **kwargs):
        super().__init__()
        # graph_inference: under torch.no_grad() the forward is replayed from a HIP graph (GraphedUpsample) -- the
        # reference's render loops call the network one image at a time, where launch gaps were a fifth of the call
        self.graph_inference = bool(graph_inference)
        self._graphed = GraphedUpsample()
        if out_dim != 3:
            raise ValueError("only out_dim = 3 (the reference's value, gaze_nerf.py:114) is supported")
        if bg_type not in ("white", "black"):
            raise ValueError("bg_type must be 'white' or 'black' (neural_renderer.py:37-52)")
        self.n_feat, self.min_feat, self.final_actvn, self.featmap_size = feat_nc, min_feat, final_actvn, featmap_size
        self.n_blocks = int(math.log2(img_size) - math.log2(featmap_size))
        ch = _channels(feat_nc, self.n_blocks, min_feat)
        self.feat_upsample_list = nn.ModuleList([_PSU(ch[i]) for i in range(self.n_blocks)])
        self.rgb_upsample = nn.Sequential(nn.Identity(), _BlurBuf())      # key "rgb_upsample.1.f" (neural_renderer.py:65-67)
        # entry 0 is built from feat_nc itself, the others from max(feat_nc >> i, min_feat) (neural_renderer.py:69-82);
        # the two differ only for feat_nc < min_feat, a configuration the reference constructs but cannot run
        self.feat_2_rgb_list = nn.ModuleList([nn.Conv2d(feat_nc if i == 0 else ch[i], 3, 1, 1, padding=0)
                                              for i in range(self.n_blocks + 1)])
        self.feat_layers = nn.ModuleList([nn.Conv2d(ch[i], ch[i + 1], 1, 1, padding=0) for i in range(self.n_blocks)])
        fill = torch.ones if bg_type == "white" else torch.zeros
        self.bg_featmap = nn.Parameter(fill((1, feat_nc, featmap_size, featmap_size), dtype=torch.float32))

    def get_bg_featmap(self):
        return self.bg_featmap

    def forward(self, x):
        params = {k: v for k, v in self.named_parameters() if k != "bg_featmap"}
        if (self.graph_inference and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and not torch.cuda.is_current_stream_capturing()):
            return self._graphed(x, params, self.n_blocks, self.min_feat, self.final_actvn)
        return neural_render(x, params, self.n_blocks, self.min_feat, self.final_actvn)

    def _apply(self, fn, *args, **kw):          # .to() / .cuda() / .half(): parameters move, captured graphs are stale
        self._graphed.clear()
        return super()._apply(fn, *args, **kw)
