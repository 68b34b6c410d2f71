"""SURVEY.md 8(f) N4, the sample -> op-input step: gazenerf_amd.data against values captured from the reference's own
GazeNerfTrainer.prepare_data / build_code_and_cam (oracle/gen_golden_prep.py, fixture g12_prepare), plus the HDF5 row
contract of datasets/eth_xgaze.py:326-352 on a synthetic row."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gazenerf_amd import data as D
from gazenerf_amd import losses as L


def _fixture_batch(g):
    para = {k[5:]: g[k] for k in g if k.startswith("para_")}
    return g["img"], g["head"], g["leye"], g["reye"], para


def test_prepare_batch_equals_the_reference_trainer():
    g = load_golden("g12_prepare")
    B, S, FM, it = [int(v) for v in g["meta"]]
    img, head, leye, reye, para = _fixture_batch(g)
    assert para["code"].dtype == torch.float64 and head.dtype == torch.uint8        # the dataset's dtypes
    pb = D.prepare_batch(img, head, leye, reye, para, base_expr_fix=g["expr_fix"], featmap_size=FM, pred_img_size=S)
    for k in ("iden", "expr", "text", "illu", "gaze", "c2w_Rmat", "c2w_Tvec", "inv_inmat", "inmat"):
        assert pb.base[k].dtype == torch.float32 and torch.equal(pb.base[k], g["base_" + k]), k
    # the fixed expression replaces the sample's own (trainer/gazenerf_trainer.py:304-310)
    assert torch.equal(pb.base["expr"], g["expr_fix"].expand(B, -1))
    assert not torch.allclose(pb.base["expr"], para["code"][:, 100:179].float())
    # focal terms and principal point scale with featmap_size / img_size; closed-form inverse
    K, Ki = pb.base["inmat"], pb.base["inv_inmat"]
    assert torch.allclose(K[:, 0, 0], (para["inmat"][:, 0, 0] * FM / S).float())
    assert float((K.bmm(Ki) - torch.eye(3)).abs().max()) <= 1e-5
    assert pb.head_mask.shape == (B, 1, S, S) and pb.full_eye_mask.shape == (B, 1, 1, 1)
    assert pb.base["c2w_Tvec"].shape == (B, 3, 1)
    # Fitter.build_code_and_cam on top of it == the reference's build_code_and_cam
    fit = L.Fitter.__new__(L.Fitter)
    fit.opt_cam = True
    for k in ("iden_offset", "expr_offset", "appea_offset", "delta_EulurAngles", "delta_Tvecs"):
        setattr(fit, k, g[k])
    shape_code, appea_code, gaze, R, T, opt_codes, delta = fit.build_code_and_cam(slice(it * B, (it + 1) * B), pb.base)
    for name, ours in (("shape_code", shape_code), ("appea_code", appea_code), ("gaze_code", gaze), ("R", R), ("T", T)):
        assert float((ours - g["out_" + name]).abs().max()) <= 1e-6, name
    assert torch.equal(opt_codes["iden"], g["iden_offset"][it * B:(it + 1) * B])


def test_prepare_batch_rejects_malformed_input():
    g = load_golden("g12_prepare")
    img, head, leye, reye, para = _fixture_batch(g)
    bad = dict(para, code=para["code"][:, :300])
    with pytest.raises(ValueError, match="code"):
        D.prepare_batch(img, head, leye, reye, bad, base_expr_fix=g["expr_fix"])
    with pytest.raises(ValueError, match="base_expr_fix"):
        D.prepare_batch(img, head, leye, reye, para, base_expr_fix=g["expr_fix"][:, :70])


def _row(rng, S=32):
    f = lambda *s: rng.standard_normal(s)
    return dict(face_patch=rng.integers(0, 256, (S, S, 3), dtype=np.uint8), head_mask=(rng.random((S, S)) < 0.7).astype(np.uint8),
                left_eye_mask=(rng.random((S, S)) < 0.1).astype(np.uint8), right_eye_mask=(rng.random((S, S)) < 0.1).astype(np.uint8),
                latent_codes=f(306), w2c_Rmat=np.eye(3), w2c_Tvec=f(3), c2w_Rmat=np.eye(3), c2w_Tvec=f(3),
                inmat=np.array([[2000.0, 0, 16], [0, 2000.0, 16], [0, 0, 1]]), inv_inmat=np.eye(3), pitchyaw_head=f(2),
                face_head_pose=f(2), facial_landmarks=f(68, 2), cam_index=np.zeros(1, np.uint8))


def test_row_contract_and_getitem_restatement():
    rng = np.random.default_rng(3)
    rows = [D.XGazeRow.from_mapping(_row(rng)) for _ in range(3)]
    row0_codes = rows[0].latent_codes
    samples = [D.row_to_sample(r, row0_codes) for r in rows]
    img, head, leye, reye, para = samples[2]
    # BGR -> RGB, u8 HWC -> float CHW in [0,1]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32
    assert float(img[0, 5, 7]) == np.float32(rows[2].face_patch[5, 7, 2]) / np.float32(255.0)
    assert float(img[2, 5, 7]) == np.float32(rows[2].face_patch[5, 7, 0]) / np.float32(255.0)
    # identity / expression / texture from row 0 of the file, illumination from the row itself (eth_xgaze.py:346-347)
    assert np.array_equal(para["code"][:279], row0_codes[:279]) and np.array_equal(para["code"][279:], rows[2].latent_codes[279:])
    assert para["code"].dtype == np.float64 and para["eye_mask"] == 0
    # two 3x3 erosions == one 5x5 minimum filter with a border that never erodes
    m = rows[2].head_mask.astype(np.int64)
    pad = np.pad(m, 2, constant_values=1)
    ref = np.min([pad[dy:dy + 32, dx:dx + 32] for dy in range(5) for dx in range(5)], axis=0)
    assert np.array_equal(head.numpy().astype(np.int64), ref) and head.dtype == torch.uint8
    bimg, bhead, bl, br, bpara = D.collate(samples)
    assert bimg.shape == (3, 3, 32, 32) and bhead.shape == (3, 32, 32) and bpara["code"].shape == (3, 306)
    assert bpara["code"].dtype == torch.float64 and bpara["c2w_Tvec"].shape == (3, 3)
    pb = D.prepare_batch(bimg, bhead, bl, br, bpara, base_expr_fix=torch.zeros(1, 79), featmap_size=4, pred_img_size=32)
    assert pb.base["iden"].shape == (3, 100) and pb.base["inv_inmat"].dtype == torch.float32
    with pytest.raises(ValueError, match="uint8"):
        D.XGazeRow.from_mapping(dict(_row(rng), head_mask=np.zeros((32, 32), np.float32)))
    with pytest.raises(ValueError, match="latent_codes"):
        D.XGazeRow.from_mapping(dict(_row(rng), latent_codes=np.zeros(300)))


@pytest.mark.gpu
def test_fitter_step_consumes_a_prepared_batch():
    """dataset batch -> prepare_batch -> Fitter.step: the reference's perform_fitting (trainer/gazenerf_trainer.py:
    478-534) from its DataLoader output onwards, on the GPU."""
    from gazenerf_amd import GazeNeRFNetAMD, synth
    dev = torch.device("cuda:0")
    g = load_golden("g12_prepare")
    B, S, FM, it = [int(v) for v in g["meta"]]
    img, head, leye, reye, para = _fixture_batch(g)
    # a camera the synthetic head is visible from: the harness's frontal intrinsics / pose at this resolution
    p = synth.synth_problem(FM, batch=B, seed=2)
    K = torch.linalg.inv(p["Kinv"].double())
    K[:, :2, :] *= S / FM
    para = dict(para, inmat=K, c2w_Rmat=p["R"].double(), c2w_Tvec=p["T"].double().squeeze(-1))
    torch.manual_seed(0)
    net = GazeNeRFNetAMD(featmap_size=FM, pred_img_size=S, num_sample_coarse=32).to(dev)
    pb = D.prepare_batch(img, head, leye, reye, para, base_expr_fix=g["expr_fix"], featmap_size=FM, pred_img_size=S, device=dev)
    assert float((pb.base["inv_inmat"].cpu() - p["Kinv"]).abs().max()) <= 1e-6
    fit = L.Fitter(net, n_rows=2 * B, lr=1e-3)
    xy = p["xy"].to(dev)
    hist = [fit.step(slice(0, B), xy, pb.base, pb.img, pb.head_mask.float(), pb.full_eye_mask, pb.left_eye_mask.float(),
                     pb.right_eye_mask.float(), t_rand=synth.synth_jitter(B, FM * FM, 32, seed=i).to(dev))["total_loss"]
            for i in range(3)]
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
