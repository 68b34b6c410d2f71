"""CPU: reference checkpoint compatibility.  tests/golden/ref_checkpoint_tiny.json
was written by oracle/gen_golden.py with the reference's own GazeNeRFNet / BaseOptions / torch.save,
laid out like trainer/gazenerf_trainer.py:156-191; the same script proved that the reference's resume
path (torch.load + net.load_state_dict) accepts a file written by save_reference_checkpoint."""
import os
import sys

import pytest
import torch

from conftest import GOLD, load_golden
from gazenerf_amd import HotPathRenderer
from gazenerf_amd import checkpoint as CK

PATH = os.path.join(GOLD, "ref_checkpoint_tiny.json")


def test_plain_torch_load_needs_the_reference_module():
    assert "configs.gazenerf_options" not in sys.modules
    with pytest.raises((ModuleNotFoundError, AttributeError)):
        torch.load(PATH, map_location="cpu", weights_only=False)


def test_load_reference_checkpoint_fills_the_renderer():
    ck = CK.load_reference_checkpoint(PATH)
    assert set(ck) >= {"net", "para", "optimizer", "iden_offset", "expr_offset", "appea_offset",
                       "delta_EulurAngles", "delta_Tvecs", "resume_epoch", "torch_random_state"}
    opt = ck["para"]
    exp = load_golden("ref_checkpoint_tiny_expect")
    assert isinstance(opt, CK.RendererOptions)
    assert (opt.num_sample_coarse, opt.featmap_nc, opt.mlp_hidden_nchannels) == (
        exp["num_sample_coarse"], exp["featmap_nc"], exp["hidden"])
    assert (opt.world_z1, opt.world_z2, opt.num_sample_fine) == (2.5, -3.5, 128)
    ren = HotPathRenderer(**CK.renderer_kwargs_from_options(opt))
    missing, ignored = CK.apply_to_renderer(ren, ck)
    assert missing == []
    assert any(k.startswith("neural_render.") for k in ignored)          # carried, not consumed
    sd = ren.state_dict()
    for k in exp:
        if k.startswith("fg_CD_predictor"):
            assert torch.equal(sd[k], exp[k]), k


def test_round_trip_keeps_every_other_entry(tmp_path):
    ck = CK.load_reference_checkpoint(PATH)
    ren = HotPathRenderer(**CK.renderer_kwargs_from_options(ck["para"]))
    CK.apply_to_renderer(ren, ck)
    with torch.no_grad():
        ren.fg_CD_predictor_eyes.density_module.weight.mul_(3.0)
    CK.update_from_renderer(ck, ren)
    out = str(tmp_path / "ckpt_5.json")
    CK.save_reference_checkpoint(out, ck)
    assert "configs.gazenerf_options" not in sys.modules and CK.RendererOptions.__module__ == "gazenerf_amd.checkpoint"
    back = CK.load_reference_checkpoint(out)
    orig = CK.load_reference_checkpoint(PATH)
    assert back["resume_epoch"] == orig["resume_epoch"] and back["para"].__dict__ == orig["para"].__dict__
    for k, v in orig["net"].items():
        if k == "fg_CD_predictor_eyes.density_module.weight":
            assert torch.equal(back["net"][k], 3.0 * v)
        else:
            assert torch.equal(back["net"][k], v), k
    assert torch.equal(back["iden_offset"], orig["iden_offset"])
    assert back["optimizer"]["param_groups"] == orig["optimizer"]["param_groups"]
    # the file names the reference's class, so its torch.load resolves its own BaseOptions
    raw = open(out, "rb").read()
    assert b"configs.gazenerf_options" in raw and b"gazenerf_amd.checkpoint" not in raw


def test_whole_network_loads_a_reference_checkpoint_strictly():
    """ckpt["net"] written by the reference's own GazeNeRFNet (oracle/gen_golden.py) -> GazeNeRFNetAMD, strict=True:
    every key (both MLPs, NeuralRenderer incl. bg_featmap and Blur's buffers) exists with the same shape."""
    from gazenerf_amd import GazeNeRFNetAMD
    ck = CK.load_reference_checkpoint(os.path.join(GOLD, "ref_checkpoint_tiny.json"))
    net = GazeNeRFNetAMD(**CK.network_kwargs_from_options(ck["para"]))
    res = net.load_state_dict(ck["net"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(net.neural_render.bg_featmap.data, ck["net"]["neural_render.bg_featmap"])
