"""NeuralRenderer + PixelShuffleUpsample + Blur.

CPU: the oracle restatement reproduces the fixtures captured from the reference's own NeuralRenderer
(oracle/gen_golden_n1.py; Blur through the kornia 0.6.4 restatement -- "unpinned", see its header).
GPU: the HIP kernels behind gnr_upsample_fwd / gnr_upsample_bwd against the same fixtures and the live oracle.
Tolerances: image 1e-4 max-abs (the north-star bound, applied to the sigmoid image); gradients rel-L2 1e-3
(exact-fp32 MFMA GEMMs; only the summation order differs from the reference).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gazenerf_amd import synth
from oracle import oracle as O


def _case(name):
    g = load_golden(name)
    cfg = {k[4:]: int(g[k]) for k in g if k.startswith("cfg_")}
    n_blocks = int(g["n_blocks"])
    params = synth.hash_renderer_params(seed=int(g["weight_seed"]), feat_nc=cfg["feat_nc"], n_blocks=n_blocks,
                                        min_feat=cfg["min_feat"], weight_scale=float(g["weight_scale"]))
    if "x" in g:
        x = g["x"]
    else:
        x = synth.synth_featmap(1, cfg["feat_nc"], cfg["featmap_size"], seed=int(g["x_seed"]))
    return g, cfg, n_blocks, params, x


def _loss(img):
    wgt = torch.linspace(0.5, 1.5, img.numel(), dtype=img.dtype, device=img.device).reshape(img.shape)
    return ((img * wgt) ** 2).mean()


def _subsample(name, t, full):
    if not full:
        return t
    if name == "out_img":
        return t[:, :, ::8, ::8]
    if name == "grad_x":
        return t[:, ::8, ::4, ::4]
    t2 = t.reshape(t.shape[0], -1)
    return t2[::16] if t2.numel() > 8192 else t2


def _rel_l2(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a.reshape(b.shape) - b).norm() / max(float(b.norm()), 1e-30))


@pytest.mark.parametrize("name", ["g8_upsampler_tiny", "g8_upsampler_full"])
def test_oracle_vs_reference_fixture(name):
    g, cfg, n_blocks, params, x = _case(name)
    full = name.endswith("full")
    if full:
        torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    xg = x.clone().requires_grad_(True)
    pg = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    img = O.neural_renderer(pg, xg, n_blocks)
    assert float((_subsample("out_img", img.detach(), full) - g["out_img"]).abs().max()) <= 1e-6
    _loss(img).backward()
    assert _rel_l2(_subsample("grad_x", xg.grad, full), g["grad_x"]) <= 1e-5
    for k, v in pg.items():
        assert _rel_l2(_subsample(k, v.grad, full), g["gradw_" + k]) <= 1e-4, k


def test_blur_matches_its_definition():
    """reflect-padded [1,2,1] x [1,2,1] / 16 (kornia filter2d, border_type='reflect', normalized=True), written out
    by hand on a small image, corners and edges included."""
    x = torch.arange(30, dtype=torch.float64).reshape(1, 1, 5, 6) ** 1.5
    got = O.blur(x)[0, 0]
    k = [1.0, 2.0, 1.0]
    refl = lambda i, n: -i if i < 0 else (2 * n - 2 - i if i >= n else i)
    for y in range(5):
        for xx in range(6):
            acc = 0.0
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    acc += k[dy + 1] * k[dx + 1] * float(x[0, 0, refl(y + dy, 5), refl(xx + dx, 6)])
            assert abs(acc / 16.0 - float(got[y, xx])) < 1e-9


def test_blur_matches_an_independent_reflect_convolution():
    """A second, independent implementation of 'reflect'-bordered correlation: scipy.ndimage.correlate(mode='mirror')
    (mirror = reflection about the edge PIXEL, the convention of torch / kornia 'reflect').  kornia itself is absent
    (Blur parity stays labelled unpinned); this at least rules out a private reading of the border rule."""
    from scipy import ndimage
    g = np.random.default_rng(3)
    x = g.standard_normal((2, 3, 9, 14))
    k = np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0]) / 16.0
    want = np.stack([np.stack([ndimage.correlate(x[b, c], k, mode="mirror") for c in range(3)]) for b in range(2)])
    got = O.blur(torch.from_numpy(x)).numpy()
    assert float(np.abs(got - want).max()) <= 1e-14


# ----------------------------------------------------------------------------- GPU
def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _run_hip(x, params, n_blocks, min_feat, dev):
    from gazenerf_amd import neural_render
    xg = x.to(dev).clone().requires_grad_(True)
    pg = {k: v.to(dev).clone().requires_grad_(True) for k, v in params.items()}
    img = neural_render(xg, pg, n_blocks=n_blocks, min_feat=min_feat)
    _loss(img).backward()
    return img.detach(), xg.grad, {k: v.grad for k, v in pg.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g8_upsampler_tiny", "g8_upsampler_full"])
def test_hip_vs_reference_fixture(name):
    dev = _dev()
    g, cfg, n_blocks, params, x = _case(name)
    full = name.endswith("full")
    img, dx, dp = _run_hip(x, params, n_blocks, cfg["min_feat"], dev)
    assert float((_subsample("out_img", img.cpu(), full) - g["out_img"]).abs().max()) <= 1e-4
    assert _rel_l2(_subsample("grad_x", dx.cpu(), full), g["grad_x"]) <= 1e-3
    for k, v in dp.items():
        assert _rel_l2(_subsample(k, v.cpu(), full), g["gradw_" + k]) <= 1e-3, k


@pytest.mark.gpu
@pytest.mark.parametrize("feat_nc,side,img,min_feat,batch", [(20, 16, 128, 6, 2), (258, 32, 64, 32, 1), (7, 32, 256, 3, 3),
                                                            (258, 16, 32, 32, 40), (12, 16, 256, 4, 1)])
def test_hip_vs_oracle_live(feat_nc, side, img, min_feat, batch):
    """Other shapes: 3 blocks from a 16x16 map, a single block, odd channel counts that are no multiple of 4; 40 stacked maps
    (3 B + 1 at B = 13: more (image, tile) pairs than one round of workgroups -- the rider shares of the weight-gradient GEMMs
    outgrew their scratch there until round 5); four blocks (GNR_UPSAMPLE_MAX_BLOCKS)."""
    dev = _dev()
    n_blocks = int(np.log2(img) - np.log2(side))
    params = synth.hash_renderer_params(seed=9, feat_nc=feat_nc, n_blocks=n_blocks, min_feat=min_feat, weight_scale=2.0)
    x = synth.synth_featmap(batch, feat_nc, side, seed=3)
    xg = x.clone().requires_grad_(True)
    pg = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.neural_renderer(pg, xg, n_blocks)
    _loss(ref).backward()
    got, dx, dp = _run_hip(x, params, n_blocks, min_feat, dev)
    assert float((got.cpu() - ref.detach()).abs().max()) <= 1e-4
    assert _rel_l2(dx.cpu(), xg.grad) <= 1e-3
    for k in pg:
        assert _rel_l2(dp[k].cpu(), pg[k].grad) <= 1e-3, k


@pytest.mark.gpu
def test_stacked_training_batch_equals_the_per_image_runs():
    """The 7 stacked maps of a B = 2 training step (3B + 1) at full size against seven single-image calls: same images and
    input gradients, weight gradients = the sum over the images.  At this size the nine split-K reductions of the backward
    (gnr_wgrad.h) do not fit the scratch at once -- the queue is flushed early and the scratch reused -- while a single image
    goes through in one batch: both orders of the queue are checked against each other."""
    dev = _dev()
    from gazenerf_amd import neural_render
    B, nb, mf = 7, 3, 32
    params = synth.hash_renderer_params(seed=11, feat_nc=258, n_blocks=nb, min_feat=mf, weight_scale=2.0)
    x = synth.synth_featmap(B, 258, 64, seed=4)

    def run(xs):
        xg = xs.to(dev).clone().requires_grad_(True)
        pg = {k: v.to(dev).clone().requires_grad_(True) for k, v in params.items()}
        img = neural_render(xg, pg, n_blocks=nb, min_feat=mf)
        w = torch.linspace(0.5, 1.5, img[0].numel(), device=dev).reshape(img[0].shape)
        (img * w).sum().backward()
        return img.detach(), xg.grad, {k: v.grad.double() for k, v in pg.items()}

    img, dx, dp = run(x)
    acc = None
    for b in range(B):
        i1, d1, p1 = run(x[b:b + 1])
        assert float((i1[0] - img[b]).abs().max()) <= 1e-6
        assert _rel_l2(d1[0].cpu(), dx[b].cpu()) <= 1e-6
        acc = p1 if acc is None else {k: acc[k] + p1[k] for k in acc}
    for k in dp:
        assert _rel_l2(dp[k].cpu().float(), acc[k].cpu().float()) <= 2e-5, k


def _variant_run(dev):
    out = {}
    for tag, (feat_nc, side, nb, mf, batch) in {"a": (258, 64, 1, 32, 1), "b": (64, 32, 2, 16, 2)}.items():
        params = synth.hash_renderer_params(seed=5, feat_nc=feat_nc, n_blocks=nb, min_feat=mf, weight_scale=2.0)
        x = synth.synth_featmap(batch, feat_nc, side, seed=2).to(dev).requires_grad_(True)
        pg = {k: v.to(dev).clone().requires_grad_(True) for k, v in params.items()}
        from gazenerf_amd import neural_render
        img = neural_render(x, pg, n_blocks=nb, min_feat=mf)
        w = torch.linspace(0.5, 1.5, img.numel(), device=dev).reshape(img.shape)
        (img * w).sum().backward()
        out[tag + "_img"] = img.detach().cpu().numpy(); out[tag + "_dx"] = x.grad.cpu().numpy()
        for k, v in pg.items():
            out[tag + "_" + k] = v.grad.cpu().numpy()
    return out


@pytest.mark.gpu
def test_every_gemm_tile_variant_gives_the_same_result():
    """conv16_plan picks one of six (row tiles, pixel tiles) instances per GEMM; gnr_set_conv16_tile (an explicit hook of the
    C ABI -- until round 3 an environment variable the library read behind the caller's back) pins one for every GEMM.
    (8,4), (11,2), (13,2) have no blur instance: pinning them also runs the separate-stencil fallback of feat_layers.  All
    must agree with the cost model's choice to rounding (different tiles sum the contraction in the same k order:
    identical up to the blur path); a pair without an instance is rejected.  Round 4: with the cost model the backward's
    du GEMM of case "b" (64 / 32 channels at 32 / 64 pixels a side) carries the un-shuffle in its epilogue
    (conv16_unshuffle_kernel; (2,8), (3,8), (4,8) pin its row tiles), with a plain pair pinned the un-shuffle is its own
    kernel behind the plain GEMM: the fused path must reproduce the two-kernel path BIT FOR BIT (same k order, same order of
    the four x.repeat-adjoint terms).  Any pinned pair also switches the chained layer_1 -> layer_2 kernel (upchain_kernel,
    case "b" block 0: 64 channels) back to two GEMMs, so the cost-model run against the pinned runs also compares the chain
    with the two-GEMM path (same products, another grouping of the contraction: rounding only)."""
    from gazenerf_amd import _lib
    lib = _lib.load()
    dev = _dev()
    res = {}
    try:
        for force in ((0, 0), (13, 2), (11, 2), (9, 2), (8, 4), (4, 4), (2, 4), (2, 8), (3, 8), (4, 8)):
            _lib.check(lib.gnr_set_conv16_tile(*force))
            res[force] = _variant_run(dev)
        assert lib.gnr_set_conv16_tile(3, 3) != 0 and b"no GEMM instance" in lib.gnr_last_error()
    finally:
        lib.gnr_set_conv16_tile(0, 0)
    base = res[(0, 0)]
    for k, v in res[(2, 4)].items():          # (MT, 8) pins only the fused un-shuffle: everything else as under (2, 4)
        if k.startswith("b_"):
            for fused in ((2, 8), (3, 8), (4, 8)):
                assert np.array_equal(res[fused][k], v), (fused, k)
    for force, got in res.items():
        for k, v in base.items():
            if k.endswith("_img"):
                assert float(np.abs(got[k] - v).max()) <= 2e-6, (force, k)
            else:
                assert float(np.linalg.norm(got[k] - v)) <= 2e-5 * float(np.linalg.norm(v)) + 1e-12, (force, k)


@pytest.mark.gpu
def test_second_backward_over_the_same_saved_state_is_identical():
    """gnr_upsample_bwd used to write d(net) over the saved pre-blur map u[i] (which its own dWf GEMM
    reads), so `backward(retain_graph=True)` followed by a second backward silently gave wrong feat_layers gradients.  d(net)
    now lives in backward scratch: two backwards over one forward are bit-identical."""
    from gazenerf_amd import neural_render
    dev = _dev()
    params = synth.hash_renderer_params(seed=7, feat_nc=64, n_blocks=2, min_feat=16, weight_scale=2.0)
    x = synth.synth_featmap(2, 64, 32, seed=3).to(dev).requires_grad_(True)
    pg = {k: v.to(dev).clone().requires_grad_(True) for k, v in params.items()}
    img = neural_render(x, pg, n_blocks=2, min_feat=16)
    w = torch.linspace(0.5, 1.5, img.numel(), device=dev).reshape(img.shape)
    loss = (img * w).sum()
    leaves = [x] + list(pg.values())
    g1 = torch.autograd.grad(loss, leaves, retain_graph=True)
    g2 = torch.autograd.grad(loss, leaves)
    for name, a, b in zip(["x"] + list(pg.keys()), g1, g2):
        assert torch.equal(a, b), name


@pytest.mark.gpu
def test_module_has_reference_state_dict_and_is_deterministic():
    from gazenerf_amd import NeuralRendererAMD
    dev = _dev()
    torch.manual_seed(0)
    net = NeuralRendererAMD(feat_nc=258, featmap_size=64, img_size=512).to(dev)
    want = set(synth.renderer_param_shapes().keys())
    have = {k.rsplit(".", 1)[0] for k in net.state_dict() if k != "bg_featmap" and not k.endswith(".f")}   # .f: Blur buffers
    assert have == want and tuple(net.bg_featmap.shape) == (1, 258, 64, 64)
    for name, (co, ci) in synth.renderer_param_shapes().items():
        assert tuple(net.state_dict()[name + ".weight"].shape) == (co, ci, 1, 1)
    x = synth.synth_featmap(1, 258, 64, seed=1).to(dev)
    a = net(x)
    a.square().mean().backward()
    g1 = {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None}
    net.zero_grad()
    b = net(x)
    b.square().mean().backward()
    assert torch.equal(a, b) and a.shape == (1, 3, 512, 512) and bool((a > 0).all()) and bool((a < 1).all())
    for k, v in net.named_parameters():
        if k != "bg_featmap":
            assert torch.equal(v.grad, g1[k]), k


@pytest.mark.gpu
def test_graph_replayed_inference_equals_the_eager_forward():
    """NeuralRendererAMD under torch.no_grad(): the forward is captured once per (shape, parameter storage) in a HIP graph
    and replayed; same bits as the eager call, in-place parameter updates are seen (the weight re-layout is inside the
    graph), a parameter moved to new storage re-captures, another batch size is another graph."""
    from gazenerf_amd import NeuralRendererAMD
    dev = _dev()
    torch.manual_seed(0)
    net = NeuralRendererAMD(feat_nc=258, featmap_size=64, img_size=512).to(dev).eval()
    x = synth.synth_featmap(1, 258, 64, seed=1).to(dev)
    eager = neural_render_eager(net, x)
    with torch.no_grad():
        a = net(x)
        b = net(x * 0.5 + 0.1)
        c = net(x)
    assert net._graphed.captures == 1 and net._graphed.replays == 3
    assert torch.equal(a, eager) and torch.equal(c, eager) and not torch.equal(a, b)
    assert a.data_ptr() != c.data_ptr()                              # outputs are copies, not the graph's buffer
    with torch.no_grad():
        net.feat_layers[1].weight.mul_(1.5)                          # in place: same storage, the replay must see it
        d = net(x)
    assert net._graphed.captures == 1 and torch.equal(d, neural_render_eager(net, x)) and not torch.equal(d, eager)
    with torch.no_grad():
        net.feat_layers[1].weight.data = net.feat_layers[1].weight.data.clone()      # new storage: captured again
        e = net(x)
        f = net(torch.cat([x, x * 0.5]))                             # B = 2: its own graph
    assert net._graphed.captures == 3 and torch.equal(e, d) and torch.equal(f[:1], d)
    y = net(x.requires_grad_(True))                                  # grad mode: the autograd op, no graph
    assert y.requires_grad and net._graphed.replays == 6 and torch.equal(y.detach(), d)


def neural_render_eager(net, x):
    from gazenerf_amd import neural_render
    with torch.no_grad():
        params = {k: v for k, v in net.named_parameters() if k != "bg_featmap"}
        return neural_render(x, params, net.n_blocks, net.min_feat, net.final_actvn)


@pytest.mark.gpu
def test_bad_inputs_are_rejected():
    from gazenerf_amd import _lib, neural_render
    dev = _dev()
    params = {k: v.to(dev) for k, v in synth.hash_renderer_params(feat_nc=12, n_blocks=1, min_feat=4).items()}
    with pytest.raises(_lib.GnrError, match="featmap_size"):
        neural_render(torch.zeros(1, 12, 8, 8, device=dev), params, n_blocks=1, min_feat=4)
    with pytest.raises(ValueError):
        neural_render(torch.zeros(1, 13, 16, 16, device=dev), params, n_blocks=1, min_feat=4)


@pytest.mark.gpu
def test_frozen_parameters_and_frozen_input_pass_null_gradient_pointers():
    """include/gnr.h documents every GnrUpsampleWeightGrads entry -- and `dw`, `d_x` themselves -- as
    "NULL == not wanted", but the register-fed weight-gradient path (wgrad16) stored through a NULL dW.  The autograd op now
    passes NULL for whatever does not require a gradient, on the reference's default network (258/129/64/32 channels:
    the 129 x 258 and 516 x 258 products take the wgrad16 path): the remaining gradients are bit-identical to the full call."""
    import ctypes as C
    from gazenerf_amd import _lib, neural_render
    from gazenerf_amd.upsample import _prep_upsample, _weights_struct, renderer_param_names
    dev = _dev()
    params = synth.hash_renderer_params(seed=5)
    x = synth.synth_featmap(2, 258, 64, seed=2).to(dev)

    def run(need_x, need):
        xg = x.clone().requires_grad_(need_x)
        pg = {k: v.to(dev).clone().requires_grad_(need(k)) for k, v in params.items()}
        img = neural_render(xg, pg)
        _loss(img).backward()
        return xg.grad, {k: v.grad for k, v in pg.items()}

    gx_all, gp_all = run(True, lambda k: True)
    gx, gp = run(True, lambda k: False)                                   # frozen renderer: d_x only, dw == NULL
    assert torch.equal(gx, gx_all) and all(v is None for v in gp.values())
    gx, gp = run(False, lambda k: True)                                   # d_x == NULL
    assert gx is None and all(torch.equal(gp[k], gp_all[k]) for k in gp_all)
    gx, gp = run(True, lambda k: k.endswith(".bias") or "layer_2" in k)   # weights NULL, their biases wanted, and a mix
    assert torch.equal(gx, gx_all)
    for k, v in gp.items():
        if k.endswith(".bias") or "layer_2" in k:
            assert torch.equal(v, gp_all[k]), k
        else:
            assert v is None, k

    # the C ABI directly: dw == NULL and d_x == NULL together is a valid (empty) request
    lib = _lib.load()
    cfg = dict(n_blocks=3, min_feat=32, final_actvn=True)
    names = renderer_param_names(3)
    flat = [params[n].to(dev) for n in names]
    p, xc, prm, _ = _prep_upsample(cfg, x, flat)
    ws = torch.empty(int(lib.gnr_upsample_workspace_bytes(C.byref(p), _lib.UP_WS_FWD)), dtype=torch.uint8, device=dev)
    sc = torch.empty(int(lib.gnr_upsample_workspace_bytes(C.byref(p), _lib.UP_WS_BWD)), dtype=torch.uint8, device=dev)
    img = torch.empty(2, 3, 512, 512, device=dev)
    w = _weights_struct(prm, 3)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.gnr_upsample_fwd(C.byref(p), C.byref(w), C.c_void_p(img.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), st), lib)
    g = torch.ones_like(img)
    _lib.check(lib.gnr_upsample_bwd(C.byref(p), C.byref(w), C.c_void_p(g.data_ptr()), None, None, C.c_void_p(ws.data_ptr()),
                                    ws.numel(), C.c_void_p(sc.data_ptr()), sc.numel(), st), lib)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_graph_cache_under_inference_mode_then_no_grad_and_module_copies():
    """a graph captured under torch.inference_mode() kept inference tensors as its static buffers and
    a later torch.no_grad() call failed at the in-place copy; deepcopy / pickle of a module that had captured raised."""
    import copy
    import io
    from gazenerf_amd import NeuralRendererAMD
    dev = _dev()
    torch.manual_seed(0)
    net = NeuralRendererAMD(feat_nc=64, featmap_size=32, img_size=128, min_feat=16).to(dev).eval()
    x = synth.synth_featmap(1, 64, 32, seed=1).to(dev)
    with torch.inference_mode():
        a = net(x)
    with torch.no_grad():
        b = net(x)
    assert net._graphed.captures == 1 and net._graphed.replays == 2 and torch.equal(a, b)
    twin = copy.deepcopy(net)
    with torch.no_grad():
        c = twin(x)
    assert twin._graphed.captures == 1 and torch.equal(c, b)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    again = torch.load(buf, weights_only=False)
    with torch.no_grad():
        assert torch.equal(again(x), b)
