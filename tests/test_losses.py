"""SURVEY.md 8(f) N4, the terms without a pretrained network: gazenerf_amd.losses against values captured from the
reference's own GazeNeRFLoss.calc_total_loss (use_vgg_loss=False) and BaseTrainer.eulurangle2Rmat
(oracle/gen_golden_n4.py), and the per-sample fitting step around the whole network on the GPU."""
import pytest
import torch

from conftest import load_golden
from gazenerf_amd import losses as L
from gazenerf_amd import synth


def _inputs(g, dev="cpu"):
    t = lambda k: g[k].to(dev)
    pred = {k: t("pred_" + k) for k in ("merge_img_face", "merge_img_eyes", "merge_img", "bg_img")}
    codes = {"bg": None, "iden": t("code_iden"), "expr": t("code_expr"), "appea": t("code_appea")}
    delta = {"delta_eulur": t("delta_eulur"), "delta_tvec": t("delta_tvec")}
    return pred, codes, delta


@pytest.mark.parametrize("use_l1", [False, True])
def test_total_loss_vs_reference_fixture(use_l1):
    g = load_golden("g10_losses")
    pred, codes, delta = _inputs(g)
    out = L.total_loss(pred, g["gt"], g["face"], g["full_eye"], g["leye"], g["reye"], codes, delta, use_l1=use_l1)
    tag = "l1_" if use_l1 else "l2_"
    keys = [k[3:] for k in g if k.startswith(tag)]
    assert "total_loss" in keys and len(keys) == 12
    for k in keys:
        assert abs(float(out[k]) - float(g[tag + k])) <= 1e-6 * max(1.0, abs(float(g[tag + k]))), k


def test_euler_to_rotation_vs_reference_fixture():
    g = load_golden("g10_losses")
    R = L.euler_to_rotation(g["euler"])
    assert float((R - g["euler_R"]).abs().max()) <= 1e-6
    eye = R.bmm(R.transpose(1, 2))
    assert float((eye - torch.eye(3)).abs().max()) <= 1e-6


@pytest.mark.gpu
def test_loss_on_gpu_matches_fixture_and_backpropagates():
    dev = torch.device("cuda:0")
    g = load_golden("g10_losses")
    pred, codes, delta = _inputs(g, dev)
    for v in pred.values():
        v.requires_grad_(True)
    out = L.total_loss(pred, g["gt"].to(dev), g["face"].to(dev), g["full_eye"].to(dev), g["leye"].to(dev),
                       g["reye"].to(dev), codes, delta)
    assert abs(float(out["total_loss"]) - float(g["l2_total_loss"])) <= 1e-5
    out["total_loss"].backward()
    assert all(torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) > 0 for v in pred.values())


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_fitter_steps_lower_the_total_loss(precision):
    """The reference's fitting step (codes + camera deltas + network, Adam with its learning-rate ratios) on an
    ETH-XGaze-shaped synthetic sample: 64x64 feature map -> 512x512 images, head / eye masks, target image."""
    from gazenerf_amd import GazeNeRFNetAMD
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    net = GazeNeRFNetAMD(featmap_size=32, pred_img_size=256, num_sample_coarse=32, precision=precision).to(dev)
    B, S = 2, 256
    p = {k: v.to(dev) for k, v in synth.synth_problem(32, batch=B, camera="5", seed=2).items()}
    base = {"iden": p["shape_code"][:, :100], "expr": p["shape_code"][:, 100:], "text": p["appea_code"][:, :100],
            "illu": p["appea_code"][:, 100:], "gaze": p["gaze"], "c2w_Rmat": p["R"], "c2w_Tvec": p["T"], "inv_inmat": p["Kinv"]}
    yy, xx = torch.meshgrid(torch.arange(S, device=dev), torch.arange(S, device=dev), indexing="ij")
    disk = lambda cy, cx, r: (((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).float().view(1, 1, S, S).expand(B, 1, S, S)
    face, leye, reye = disk(128, 128, 96), disk(104, 88, 18), disk(104, 168, 18)
    full_eye = torch.clamp(leye + reye, 0, 1)
    gt = torch.rand(B, 3, S, S, device=dev) * face + (1.0 - face)
    fit = L.Fitter(net, n_rows=4, lr=1e-3)
    hist = []
    for i in range(4):
        t_rand = synth.synth_jitter(B, 32 * 32, 32, seed=i).to(dev)
        hist.append(fit.step(slice(0, B), p["xy"], base, gt, face, full_eye, leye, reye, t_rand=t_rand)["total_loss"])
    assert hist[-1] < hist[0], hist
    assert float(fit.iden_offset.abs().sum()) > 0 and float(fit.delta_Tvecs.abs().sum()) > 0      # they moved
    assert float(fit.iden_offset[2:].abs().sum()) == 0                                           # untouched rows did not
