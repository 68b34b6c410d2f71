"""Feature-map merge: the step right after the volumetric hot path.

``merge_featmaps`` replaces models/gaze_nerf.py:175-203 (background blend with
``NeuralRenderer.bg_featmap``, rotation of the eye-stream feature triplets by the gaze through
``rotate`` / ``rotation_matrix_2d`` (utils/model_utils.py:11-46), ``torch.maximum``) with one HIP
kernel each for forward and backward behind ``gnr_merge_fwd`` / ``gnr_merge_bwd``.  It consumes the
four outputs of ``render_two_stream`` in their native [B, C, N_r] layout and returns the three
feature maps the reference hands to ``NeuralRenderer``.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .render import _check_tensor, _stream_ptr


def _problem(ff, af, fe, ae, bg, gaze):
    B, Cn = ff.shape[0], ff.shape[1]
    n_pix = ff[0, 0].numel()
    for name, t, shp in (("feat_face", ff, None), ("bg_alpha_face", af, None), ("feat_eyes", fe, None),
                         ("bg_alpha_eyes", ae, None), ("bg_featmap", bg, None), ("gaze", gaze, (B, 2))):
        _check_tensor(name, t, shp)
    if fe.shape != ff.shape or af.numel() != B * n_pix or ae.numel() != B * n_pix or bg.numel() != Cn * n_pix:
        raise ValueError("merge_featmaps: inconsistent shapes")
    ts = [t.contiguous() for t in (ff, af, fe, ae, bg, gaze)]
    p = _lib.GnrMergeProblem()
    p.batch, p.n_pix, p.feat_nc = B, n_pix, Cn
    (p.feat_face, p.bg_alpha_face, p.feat_eyes, p.bg_alpha_eyes, p.bg_featmap, p.gaze) = [t.data_ptr() for t in ts]
    return p, ts


class _MergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ff, af, fe, ae, bg, gaze):
        lib = _lib.load()
        p, ts = _problem(ff, af, fe, ae, bg, gaze)
        outs = [torch.empty_like(ts[0]) for _ in range(3)]
        with torch.cuda.device(ff.device):
            _lib.check(lib.gnr_merge_fwd(C.byref(p), *[C.c_void_p(o.data_ptr()) for o in outs], _stream_ptr(ff.device)), lib)
        ctx.save_for_backward(*ts)
        ctx.shapes = [t.shape for t in (ff, af, fe, ae, bg, gaze)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_mf, g_ep, g_m):
        lib = _lib.load()
        ts = list(ctx.saved_tensors)
        p, ts = _problem(*ts)
        gs = [g.contiguous() if g is not None else None for g in (g_mf, g_ep, g_m)]
        d = [torch.empty_like(t) for t in ts]
        dev = ts[0].device
        with torch.cuda.device(dev):
            scratch = torch.empty(max(int(lib.gnr_merge_scratch_bytes(C.byref(p))), 256), dtype=torch.uint8, device=dev)
            rc = lib.gnr_merge_bwd(C.byref(p), *[C.c_void_p(g.data_ptr()) if g is not None else None for g in gs],
                                   *[C.c_void_p(t.data_ptr()) for t in d],
                                   C.c_void_p(scratch.data_ptr()), scratch.numel(), _stream_ptr(dev))
            _lib.check(rc, lib)
        return tuple(t.reshape(s) for t, s in zip(d, ctx.shapes))


def merge_featmaps(feat_face, bg_alpha_face, feat_eyes, bg_alpha_eyes, bg_featmap, gaze):
    """[B,C,N_r] (or [B,C,H,W]) maps, bg_alpha [B,1,...], bg_featmap [1,C,...], gaze [B,2]
    -> (merge_featmap_face, eyes_planes, merge_featmap), same shape as feat_face."""
    return _MergeFn.apply(feat_face, bg_alpha_face, feat_eyes, bg_alpha_eyes, bg_featmap, gaze)
