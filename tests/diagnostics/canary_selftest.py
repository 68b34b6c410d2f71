"""GPU diagnostic for a -DGNR_CANARY build (tools/session.sh <name> canary; gazenerf_amd/csrc/gnr_canary.h).

    GNR_ALLOW_EXPERIMENTAL_LIB=1 python tests/diagnostics/canary_selftest.py expect-clean | expect-hit

expect-clean (-DGNR_CANARY=1): a small forward + backward of the hot path and of the upsampler must pass with every carve-internal gap
intact.  expect-hit (-DGNR_CANARY=2: the column-sum shares of every weight-gradient GEMM start 64 floats late, so they end inside the
gap behind them -- the round-5 overrun in miniature): the backward MUST fail with the canary's message."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gazenerf_amd import _lib, neural_render, render, synth        # noqa: E402


def main():
    want_hit = sys.argv[1] == "expect-hit"
    info = _lib.build_info()
    assert "GNR_CANARY" in info, "not a canary build: %s" % info
    dev = torch.device("cuda:0")
    hits = []
    # hot path
    p = {k: v.to(dev) for k, v in synth.synth_problem(64, batch=3, camera="3", seed=1, ray_subset=torch.arange(40) * 7 % 4096).items()}
    face = {k: v.to(dev).requires_grad_(True) for k, v in synth.hash_mlp_params("face", seed=0, density_scale=10.0).items()}
    eyes = {k: v.to(dev).requires_grad_(True) for k, v in synth.hash_mlp_params("eyes", seed=0, density_scale=10.0).items()}
    try:
        out = render.render_two_stream(p["xy"], p["R"], p["T"], p["Kinv"], p["shape_code"], p["gaze"], p["appea_code"], face, eyes,
                                       n_samples=32, t_rand=synth.synth_jitter(3, 40, 32, seed=1).to(dev))
        sum((out["feat_" + t] ** 2).mean() + out["bg_alpha_" + t].mean() for t in ("face", "eyes")).backward()
        torch.cuda.synchronize()
    except _lib.GnrError as e:
        hits.append("hot path: %s" % e)
    # upsampler
    params = {k: v.to(dev).requires_grad_(True) for k, v in synth.hash_renderer_params(seed=3, feat_nc=64, n_blocks=2, min_feat=16).items()}
    x = synth.synth_featmap(3, 64, 16, seed=2).to(dev).requires_grad_(True)
    try:
        neural_render(x, params, n_blocks=2, min_feat=16).sum().backward()
        torch.cuda.synchronize()
    except _lib.GnrError as e:
        hits.append("upsampler: %s" % e)
    for h in hits:
        print(h)
    ok = (len(hits) == 2 and all("canary" in h for h in hits)) if want_hit else not hits
    print("canary selftest (%s, %s): %s" % (sys.argv[1], info, "OK" if ok else "FAILED"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
