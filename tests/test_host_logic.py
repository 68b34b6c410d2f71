"""CPU: host-side logic that needs no GPU -- parameter ordering/shape checks of the render op, the
module's reference-compatible state dict, the synthetic recipe against SURVEY.md 8(d) constants."""
import pytest
import torch

from gazenerf_amd import HotPathRenderer, render, synth


def test_param_order_and_shapes():
    p = synth.hash_mlp_params("face")
    lst = render.params_to_list(p)
    assert [tuple(t.shape) for t in lst[:4]] == [(384, 244, 1, 1), (384,), (384, 384, 1, 1), (384,)]
    assert tuple(lst[10].shape) == (384, 628, 1, 1)            # FeaExt_module_5: skip layer
    assert tuple(lst[16].shape) == (1, 384, 1, 1) and float(lst[17].abs().sum()) == 0.0   # density bias 0
    assert tuple(lst[20].shape) == (192, 511, 1, 1) and tuple(lst[22].shape) == (258, 192, 1, 1)
    with pytest.raises(ValueError):
        render.params_to_list(lst[:-1])
    shapes = render._expected_param_shapes(384, 244, 127, 258)
    assert len(shapes) == 24 and shapes[10] == (384, 628)


def test_module_state_dict_matches_reference_names():
    net = HotPathRenderer(hier_sampling=True)
    sd = net.state_dict()
    for pre in ("fg_CD_predictor_face", "fg_CD_predictor_eyes", "fine_fg_CD_predictor"):
        for k in render.PARAM_ORDER:
            assert "%s.%s" % (pre, k) in sd
    n = sum(v.numel() for k, v in sd.items() if k.startswith("fg_CD_predictor_face."))
    assert n == 1518979
    assert float(sd["fg_CD_predictor_face.density_module.bias"].abs().sum()) == 0.0
    with pytest.raises(RuntimeError):
        net.fg_CD_predictor_face(torch.zeros(1))


def test_synth_matches_survey_constants():
    k = synth.scaled_kinv(64)[0]
    assert abs(float(k[0, 0]) - 0.007790804840624332 / 2) < 1e-9
    assert abs(float(k[0, 2]) + 0.12553827464580536) < 1e-9
    r, t = synth.orbit_camera(3)
    ref = torch.tensor([[0.92791, 0.06257, -0.36751], [0.0, -0.98582, -0.16783], [-0.37279, 0.15574, -0.91475]])
    assert float((r[0] - ref).abs().max()) < 1e-4
    assert float((t.flatten() - torch.tensor([4.82105, 2.20170, 12.0])).abs().max()) < 1e-4
    xy = synth.pixel_grid(64)
    assert xy.shape == (1, 2, 4096) and float(xy[0, 0, 65]) == 1.0 and float(xy[0, 1, 65]) == 1.0
    s, a, g = synth.synth_codes(3, seed=1)
    assert s.shape == (3, 179) and a.shape == (3, 127) and g.shape == (3, 2) and float(g.abs().max()) <= 0.5


def test_bench_flop_constant_matches_survey():
    import bench
    # L0 63*384, L1-4 4*384^2, L5 (63+384)*384, L6-7 2*384^2, sigma 384, RGB0 384^2,
    # RGB1 384*192, RGB2 192*258
    macs = 63 * 384 + 4 * 384 * 384 + (63 + 384) * 384 + 2 * 384 * 384 + 384 + 384 * 384 + 384 * 192 + 192 * 258
    assert macs == 1351680 and bench.FLOP_PER_SAMPLE_STREAM == 2 * macs


def test_packed_weight_cache_invalidation():
    """render.PackedWeightCache (round-2 advice): a reallocated workspace must forget its key at once (a call that
    raises before store() must not leave the OLD key describing the NEW buffer), and the key includes data_ptr."""
    import torch
    from gazenerf_amd import render
    c = render.PackedWeightCache()
    params = [torch.zeros(4), torch.zeros(3)]
    k1, k2 = ("fp32", 1, torch.device("cpu")), ("fp32", 2, torch.device("cpu"))
    ws, hit = c.lookup(256, k1, params)
    assert not hit
    c.store(k1, params)
    assert c.lookup(256, k1, params) == (ws, True)
    ws2, hit = c.lookup(512, k2, params)          # other shape: new buffer ...
    assert not hit and ws2 is not ws
    # ... and the call "raised" before store(): the next call with the FIRST shape must not hit on the new buffer
    ws3, hit = c.lookup(256, k1, params)
    assert not hit
    c.store(k1, params)
    assert c.lookup(256, k1, params)[1]
    params[0].add_(1.0)                           # in-place write bumps _version
    assert not c.lookup(256, k1, params)[1]
    c.store(k1, params)
    params[1].data = torch.ones(3)                # .data swap keeps _version, moves data_ptr
    assert not c.lookup(256, k1, params)[1]


def test_hot_path_renderer_drops_its_caches_when_parameters_can_change():
    import torch
    from gazenerf_amd import HotPathRenderer
    net = HotPathRenderer()
    for poke in (lambda: net.train(), lambda: net.eval(), lambda: net.float(),
                 lambda: net.load_state_dict(net.state_dict())):
        net._wcache.shape_key, net._wcache.ws = ("x",), torch.zeros(1)
        poke()
        assert net._wcache.ws is None and net._wcache.shape_key is None
