"""Reference checkpoint compatibility.

The reference trainer checkpoints with ``torch.save`` of one dict
(trainer/gazenerf_trainer.py:156-191): RNG states, epoch info, ``"net"`` = ``GazeNeRFNet.state_dict()``,
``"para"`` = the live ``BaseOptions`` object (pickled by class reference
``configs.gazenerf_options.BaseOptions``), the optimizer state and the five per-row offset tables;
``--resume`` reads it back with ``torch.load`` + ``net.load_state_dict(check_dict["net"])``
(trainer/gazenerf_trainer.py:99-120).  Two things make that file awkward outside the reference tree:

* unpickling ``"para"`` needs the module ``configs.gazenerf_options`` -- absent here;
* a file written here must again resolve to that class when the reference loads it.

``load_reference_checkpoint`` unpickles with a class map (``BaseOptions`` -> ``RendererOptions``, a
plain attribute bag); ``save_reference_checkpoint`` pickles ``RendererOptions`` under the reference's
module path.  ``apply_to_renderer`` / ``update_from_renderer`` move the hot-path parameters
(``fg_CD_predictor_{face,eyes}.*``, ``fine_fg_CD_predictor.*``) between the ``"net"`` entry and a
``HotPathRenderer``; every other entry (``neural_render.*``, optimizer, offsets, RNG) is carried
through untouched so a checkpoint survives a round trip through this package.
"""
from __future__ import annotations

import pickle
import sys
import types
from typing import Dict, Tuple

import torch

REF_OPTIONS_MODULE = "configs.gazenerf_options"
REF_OPTIONS_CLASS = "BaseOptions"
HOT_PATH_PREFIXES = ("fg_CD_predictor_face.", "fg_CD_predictor_eyes.", "fine_fg_CD_predictor.")


class RendererOptions(object):
    """Stand-in for the reference's ``BaseOptions`` (configs/gazenerf_options.py:1-35): same attribute
    names (num_sample_coarse, num_sample_fine, world_z1, world_z2, mlp_hidden_nchannels, featmap_size,
    featmap_nc, pred_img_size, *_code_dims ...), no behaviour."""

    def __repr__(self):
        return "RendererOptions(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.__dict__.items()))


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == REF_OPTIONS_MODULE and name == REF_OPTIONS_CLASS:
            return RendererOptions
        return super().find_class(module, name)


def _pickle_shim():
    shim = types.ModuleType("gazenerf_amd._ckpt_pickle")
    shim.Unpickler = _Unpickler
    shim.load = lambda f, **kw: _Unpickler(f, **kw).load()
    shim.__name__ = "pickle"
    for k in ("Pickler", "dump", "dumps", "loads", "HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL", "PickleError",
              "PicklingError", "UnpicklingError"):
        setattr(shim, k, getattr(pickle, k))
    return shim


def load_reference_checkpoint(path, map_location="cpu") -> Dict:
    """torch.load of a reference checkpoint without the reference on sys.path."""
    return torch.load(path, map_location=map_location, pickle_module=_pickle_shim(), weights_only=False)


def apply_to_renderer(renderer, ckpt: Dict) -> Tuple[list, list]:
    """Fill a HotPathRenderer from ckpt["net"]; returns (missing_hot_path_keys, ignored_other_keys)."""
    net = ckpt["net"]
    own = renderer.state_dict()
    missing = [k for k in own if k not in net]
    ignored = [k for k in net if k not in own]
    renderer.load_state_dict({k: v for k, v in net.items() if k in own}, strict=False)
    return missing, ignored


def renderer_kwargs_from_options(opt) -> Dict:
    """HotPathRenderer(**kwargs) matching a checkpoint's ``para`` entry."""
    return dict(num_sample_coarse=opt.num_sample_coarse, num_sample_fine=opt.num_sample_fine,
                world_z1=opt.world_z1, world_z2=opt.world_z2, hidden=opt.mlp_hidden_nchannels,
                featmap_nc=opt.featmap_nc, shape_dims=opt.iden_code_dims + opt.expr_code_dims,
                gaze_dims=opt.eye_code_dims, appea_dims=opt.text_code_dims + opt.illu_code_dims)


def network_kwargs_from_options(opt) -> Dict:
    """GazeNeRFNetAMD(**kwargs) matching a checkpoint's ``para`` entry (models/gaze_nerf.py:40-46, 111-119)."""
    kw = renderer_kwargs_from_options(opt)
    kw.update(featmap_size=opt.featmap_size, pred_img_size=opt.pred_img_size, bg_type=getattr(opt, "bg_type", "white"))
    return kw


def update_from_renderer(ckpt: Dict, renderer) -> Dict:
    """Write the renderer's parameters back into ckpt["net"] (other entries untouched)."""
    net = ckpt["net"]
    for k, v in renderer.state_dict().items():
        net[k] = v.detach().cpu().clone()
    return ckpt


def save_reference_checkpoint(path, ckpt: Dict) -> None:
    """torch.save in the reference's format: a RendererOptions under "para" is pickled as
    configs.gazenerf_options.BaseOptions so the reference's torch.load resolves its own class."""
    cls = type(ckpt.get("para"))
    patched = cls is RendererOptions
    saved_mods = {}
    old = (RendererOptions.__module__, RendererOptions.__qualname__, RendererOptions.__name__)
    try:
        if patched:
            pkg_name = REF_OPTIONS_MODULE.split(".")[0]
            for name in (pkg_name, REF_OPTIONS_MODULE):
                saved_mods[name] = sys.modules.get(name)
            pkg = types.ModuleType(pkg_name)
            mod = types.ModuleType(REF_OPTIONS_MODULE)
            setattr(mod, REF_OPTIONS_CLASS, RendererOptions)
            pkg.gazenerf_options = mod
            sys.modules[pkg_name], sys.modules[REF_OPTIONS_MODULE] = pkg, mod
            RendererOptions.__module__ = REF_OPTIONS_MODULE
            RendererOptions.__qualname__ = RendererOptions.__name__ = REF_OPTIONS_CLASS
        torch.save(ckpt, path)
    finally:
        if patched:
            RendererOptions.__module__, RendererOptions.__qualname__, RendererOptions.__name__ = old
            for name, m in saved_mods.items():
                if m is None:
                    sys.modules.pop(name, None)
                else:
                    sys.modules[name] = m
